// storage.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// Row gather / key patch scatter for the host tie-break, key hex text, bucket histogram + scatter (bucket CSR), the
// kernel-driven device -> pinned-host copy.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
__global__ void gather_rows_kernel(const float* __restrict__ X, int64_t ldx, int dim, const int64_t* __restrict__ rows,
                                   int64_t m, float* __restrict__ dst) {
  const int64_t t = blockIdx.x;
  if (t >= m) return;
  const float* src = X + rows[t] * ldx;
  float* out = dst + t * (int64_t)dim;
  for (int k = threadIdx.x; k < dim; k += blockDim.x) out[k] = src[k];
}

// Copy the X row of every tie entry (count read on the device) into a staging matrix, so the host can
// fetch entries and their vectors without a round trip in between.
__global__ void gather_tied_rows_kernel(const float* __restrict__ X, int64_t ldx, int dim,
                                        const int64_t* __restrict__ tie_list, const int32_t* __restrict__ tie_count,
                                        int tie_cap, float* __restrict__ dst) {
  const int cnt = min(*tie_count, tie_cap);
  for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
    const float* src = X + (tie_list[2 * (int64_t)e] >> 16) * ldx;
    float* out = dst + (int64_t)e * dim;
    for (int k = threadIdx.x; k < dim; k += blockDim.x) out[k] = src[k];
  }
}

// Lower-case hex of every key byte (what `bytes.hex()` gives; the text of the reference's bucket keys,
// lshrs/storage/redis.py:225), 16 input bytes -> 32 output characters per thread.
// Device memory -> page-locked host memory by the CUs instead of a copy engine (lshrs_copy_to_host_u8): 16 bytes per lane,
// grid-stride.  `head` bytes in front of the first 16-byte boundary and the `tail` bytes behind the last whole 16 go one by
// one (both < 16 when source and destination sit alike modulo 16; the caller passes everything as `tail` when they do not -
// then every workgroup takes its share of the bytes).
__global__ __launch_bounds__(256) void copy_to_host_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int head,
                                                            int64_t n16, int64_t tail) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t me = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* __restrict__ s16 = reinterpret_cast<const uint4*>(src + head);
  uint4* __restrict__ d16 = reinterpret_cast<uint4*>(dst + head);
  for (int64_t i = me; i < n16; i += stride) d16[i] = s16[i];
  if (me < head) dst[me] = src[me];
  const int64_t t0 = head + 16 * n16;
  for (int64_t i = me; i < tail; i += stride) dst[t0 + i] = src[t0 + i];
}

__global__ void keys_to_hex_kernel(const uint8_t* __restrict__ keys, int64_t nbytes, uint8_t* __restrict__ hex) {
  const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (t >= nbytes) return;
  const int64_t end = t + 16 < nbytes ? t + 16 : nbytes;
  for (int64_t i = t; i < end; ++i) {
    const uint8_t b = keys[i];
    const uint8_t hi = b >> 4, lo = b & 15;
    hex[2 * i] = (uint8_t)(hi < 10 ? '0' + hi : 'a' + (hi - 10));
    hex[2 * i + 1] = (uint8_t)(lo < 10 ? '0' + lo : 'a' + (lo - 10));
  }
}

// Storage-op path: the (band, key) buckets of a batch as a CSR, by a counting sort per band.  One thread
// per vector: its key row is one contiguous read, its num_bands bucket indices go through atomics on a table of
// num_bands << (8 * band_bytes) bins (4 MB at 16 bands x 16-bit keys: L2-resident).  bin = band << (8 B) | key, the key
// read little-endian (key bytes = bin & 0xFF, (bin >> 8) & 0xFF).
template <int BB>
__global__ void bucket_histogram_kernel(const uint8_t* __restrict__ keys, int64_t n, int num_bands,
                                        int32_t* __restrict__ counts) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint8_t* k = keys + row * (int64_t)num_bands * BB;
  for (int b = 0; b < num_bands; ++b) {
    const unsigned key = BB == 1 ? (unsigned)k[b] : ((unsigned)k[2 * b] | ((unsigned)k[2 * b + 1] << 8));
    atomicAdd(counts + (((size_t)b << (8 * BB)) | key), 1);
  }
}

// members[offsets[bin] + (arrival order within the bin)] = ids[row]: the order inside a bucket is unspecified (the
// buckets are sets - SADD, lshrs/storage/redis.py:408-416).
template <int BB>
__global__ void bucket_scatter_kernel(const uint8_t* __restrict__ keys, const int64_t* __restrict__ ids, int64_t n,
                                      int num_bands, const int64_t* __restrict__ offsets, int32_t* __restrict__ cursors,
                                      int64_t* __restrict__ members) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const uint8_t* k = keys + row * (int64_t)num_bands * BB;
  const int64_t id = ids[row];
  for (int b = 0; b < num_bands; ++b) {
    const unsigned key = BB == 1 ? (unsigned)k[b] : ((unsigned)k[2 * b] | ((unsigned)k[2 * b + 1] << 8));
    const size_t bin = ((size_t)b << (8 * BB)) | key;
    const int pos = atomicAdd(cursors + bin, 1);
    members[offsets[bin] + pos] = id;
  }
}

__global__ void scatter_keys_kernel(uint8_t* __restrict__ keys, int num_bands, int bb, const int64_t* __restrict__ rows,
                                    const int32_t* __restrict__ bands, const uint8_t* __restrict__ patch, int64_t m) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * bb) return;
  const int64_t e = t / bb;
  const int byte = (int)(t % bb);
  keys[(rows[e] * num_bands + bands[e]) * (int64_t)bb + byte] = patch[e * bb + byte];
}
}  // namespace

extern "C" {

int lshrs_gather_rows_f32(const float* X, int64_t ldx, int32_t dim, const int64_t* rows, int64_t m, float* dst,
                          void* stream) {
  if (m == 0) return 0;
  if (X == nullptr || rows == nullptr || dst == nullptr || dim <= 0 || m < 0 || ldx < dim) return LSHRS_E_BADARG;
  if (m > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)m), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx, dim,
                     rows, m, dst);
  return -(int)hipGetLastError();
}

int lshrs_gather_tied_rows_f32(const float* X, int64_t ldx, int32_t dim, const int64_t* tie_list,
                               const int32_t* tie_count, int32_t tie_cap, float* dst, void* stream) {
  if (tie_cap == 0) return 0;
  if (X == nullptr || tie_list == nullptr || tie_count == nullptr || dst == nullptr || dim <= 0 || tie_cap < 0 ||
      ldx < dim)
    return LSHRS_E_BADARG;
  const int blocks = tie_cap < 2048 ? tie_cap : 2048;
  hipLaunchKernelGGL(gather_tied_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), X,
                     ldx, dim, tie_list, tie_count, tie_cap, dst);
  return -(int)hipGetLastError();
}

int lshrs_copy_to_host_u8(const void* src, void* dst_host, int64_t nbytes, void* stream) {
  if (nbytes == 0) return 0;
  if (src == nullptr || dst_host == nullptr || nbytes < 0) return LSHRS_E_BADARG;
  void* dst = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&dst, dst_host, 0);        // (page-locked, device-visible: else an error, not a fault)
  if (e != hipSuccess || dst == nullptr) return e != hipSuccess ? -(int)e : LSHRS_E_BADARG;
  // 16 bytes per lane wherever source and destination sit alike modulo 16 (a row-offset slice of a flag buffer, an odd
  // length: up to 15 bytes in front of the first boundary and behind the last go one by one); where they do not, bytes -
  // shared out over the whole grid, not left to one workgroup
  const uintptr_t sa = reinterpret_cast<uintptr_t>(src), da = reinterpret_cast<uintptr_t>(dst);
  int head = 0;
  int64_t n16 = 0, tail = nbytes;
  if (((sa ^ da) & 15) == 0) {
    head = (int)((16 - (sa & 15)) & 15);
    if (head > nbytes) head = (int)nbytes;
    n16 = (nbytes - head) / 16;
    tail = nbytes - head - 16 * n16;
  }
  const int64_t work = n16 > tail ? n16 : tail;
  const int64_t want = (work + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
  hipLaunchKernelGGL(copy_to_host_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), head, n16, tail);
  return -(int)hipGetLastError();
}

int lshrs_keys_to_hex_u8(const uint8_t* keys, int64_t nbytes, uint8_t* hex, void* stream) {
  if (nbytes == 0) return 0;
  if (keys == nullptr || hex == nullptr || nbytes < 0) return LSHRS_E_BADARG;
  const int64_t threads = (nbytes + 15) / 16;
  if ((threads + 255) / 256 > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  hipLaunchKernelGGL(keys_to_hex_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), keys, nbytes, hex);
  return -(int)hipGetLastError();
}

int lshrs_bucket_histogram_u8(const uint8_t* keys, int64_t n, int32_t num_bands, int32_t band_bytes, int32_t* counts,
                              void* stream) {
  if (n == 0) return 0;
  if (keys == nullptr || counts == nullptr || n < 0 || num_bands <= 0) return LSHRS_E_BADARG;
  if (band_bytes < 1 || band_bytes > 2 || num_bands > 32768 || (n + 255) / 256 > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (band_bytes == 1)
    hipLaunchKernelGGL(bucket_histogram_kernel<1>, grid, block, 0, s, keys, n, num_bands, counts);
  else
    hipLaunchKernelGGL(bucket_histogram_kernel<2>, grid, block, 0, s, keys, n, num_bands, counts);
  return -(int)hipGetLastError();
}

int lshrs_bucket_scatter_u8(const uint8_t* keys, const int64_t* ids, int64_t n, int32_t num_bands, int32_t band_bytes,
                            const int64_t* offsets, int32_t* cursors, int64_t* members, void* stream) {
  if (n == 0) return 0;
  if (keys == nullptr || ids == nullptr || offsets == nullptr || cursors == nullptr || members == nullptr || n < 0 ||
      num_bands <= 0)
    return LSHRS_E_BADARG;
  if (band_bytes < 1 || band_bytes > 2 || num_bands > 32768 || (n + 255) / 256 > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (band_bytes == 1)
    hipLaunchKernelGGL(bucket_scatter_kernel<1>, grid, block, 0, s, keys, ids, n, num_bands, offsets, cursors, members);
  else
    hipLaunchKernelGGL(bucket_scatter_kernel<2>, grid, block, 0, s, keys, ids, n, num_bands, offsets, cursors, members);
  return -(int)hipGetLastError();
}

int lshrs_scatter_band_keys_u8(uint8_t* keys, int32_t num_bands, int32_t band_bytes, const int64_t* rows,
                               const int32_t* bands, const uint8_t* patch, int64_t m, void* stream) {
  if (m == 0) return 0;
  if (keys == nullptr || rows == nullptr || bands == nullptr || patch == nullptr || num_bands <= 0 || band_bytes <= 0 ||
      m < 0)
    return LSHRS_E_BADARG;
  const int64_t total = m * band_bytes;
  if ((total + 255) / 256 > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  hipLaunchKernelGGL(scatter_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), keys, num_bands, band_bytes, rows, bands, patch, m);
  return -(int)hipGetLastError();
}

}  // extern "C"
