// mfma_probe.hip — feeds hand-made operands to v_mfma_f32_16x16x32_{f16,bf16} and returns the raw results, so that the
// accumulation model of the instruction (exactness of the products, where and how the sum is rounded, subnormal
// handling) can be worked out offline (tools/probes/mfma_model.py).  Diagnostic only; not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave = 16 tests: row i of A is test i's a[32], column j of B is test j's b[32]; D[i][i] is test i's result.
// Operand layout: lane l holds row/col (l & 15), k = 8 (l >> 4) + 0..7.  Accumulator: col = l & 15, row = 4 (l >> 4) + reg.
template <int KIND>
__global__ void probe_kernel(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n) {
  const int lane = threadIdx.x & 63;
  const int t0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
  const int rc = lane & 15, kg = lane >> 4;
  const int ta = t0 + rc < n ? t0 + rc : n - 1;
  u16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(size_t)ta * 32 + 8 * kg + j];
    b[j] = B[(size_t)ta * 32 + 8 * kg + j];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * kg + r;
    if (row == rc && t0 + row < n) c[r] = C[t0 + row];
  }
  f32x4 d;
  if (KIND == 0)
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * kg + r;
    if (row == rc && t0 + row < n) D[t0 + row] = d[r];
  }
}

extern "C" int mfma_probe_run(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n, int kind) {
  uint16_t *dA = nullptr, *dB = nullptr;
  float *dC = nullptr, *dD = nullptr;
  if (hipMalloc(&dA, (size_t)n * 64) != hipSuccess || hipMalloc(&dB, (size_t)n * 64) != hipSuccess ||
      hipMalloc(&dC, (size_t)n * 4) != hipSuccess || hipMalloc(&dD, (size_t)n * 4) != hipSuccess)
    return -1;
  (void)hipMemcpy(dA, A, (size_t)n * 64, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B, (size_t)n * 64, hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, C, (size_t)n * 4, hipMemcpyHostToDevice);
  const int waves = (n + 15) / 16, blocks = (waves + 3) / 4;
  if (kind == 0)
    hipLaunchKernelGGL(probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, dD, n);
  else
    hipLaunchKernelGGL(probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, dD, n);
  const hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(D, dD, (size_t)n * 4, hipMemcpyDeviceToHost);
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
  return e == hipSuccess ? 0 : -(int)e;
}
