// Does the 16-byte LDS-DMA (global_load_lds_dwordx4, gfx950) take a global address that is only 4-byte aligned?
// hipcc --offload-arch=gfx950 -O2 -o lds_dma_align_probe tools/probes/lds_dma_align_probe.hip && ./lds_dma_align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
__global__ void k(const float* x, float* y, int off) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(x + off + 7 * lane), (LDS_AS void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) y[4 * lane + e] = lds[4 * lane + e];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *x, *y;
  hipMalloc(&x, 4096 * 4); hipMalloc(&y, 256 * 4);
  hipMemcpy(x, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  int bad = 0;
  for (int off = 0; off < 4; ++off) {
    hipMemset(y, 0, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, off);
    std::vector<float> o(256);
    hipError_t e = hipMemcpy(o.data(), y, 256 * 4, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) if (o[4 * l + q] != (float)(off + 7 * l + q)) ++wrong;
    printf("offset %d floats: err=%d wrong=%d first lane got %g %g %g %g (want %d..)\n", off, (int)e, wrong, o[0], o[1], o[2], o[3], off);
    bad += wrong;
  }
  printf(bad ? "LDS-DMA needs aligned addresses\n" : "LDS-DMA takes 4-byte aligned global addresses\n");
  return 0;
}
