// rerank.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// K2 cosine of gathered candidates (gather + dot + norm), L2 normalisation, K3 per-query descending order (LDS bitonic
// network; longer lists through global memory).
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
// ------------------------------------------------------------------------------------------
// K2: cosine of gathered candidates against a query.  One workgroup = one (query, slice of
// its candidates); the query sits in LDS, each wave streams whole candidate rows (16 B per
// lane per load, four rows in flight), reduces dot and ||c||^2 across the wave, and lane 0
// writes dot / (||c|| * ||q||).
// ------------------------------------------------------------------------------------------
constexpr int kCosThreads = 256;
constexpr int kCosWaves = kCosThreads / 64;
constexpr int kCosInflight = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// RAGGED (ABI 7, lshrs_cosine_ragged_f32): query qi has row_cnt[qi] candidates, listed - and scored - at row_off[qi] of the
// flat cand_idx / scores arrays (the candidate lists of a batch of LSH queries, lshrs/core/main.py:629-646); what is wrong with
// a candidate is OR-ed into err[0] (1 zero norm, 2 index outside the corpus) instead of a status byte per candidate.
template <bool ALIGNED, bool RAGGED>
__global__ __launch_bounds__(kCosThreads) void cosine_kernel(const float* __restrict__ corpus, int64_t m, int64_t ldc,
                                                             int dim, const float* __restrict__ queries,
                                                             const int64_t* __restrict__ cand_idx, int c, int slices,
                                                             float* __restrict__ scores, uint8_t* __restrict__ status,
                                                             uint8_t* __restrict__ qstatus,
                                                             const int64_t* __restrict__ row_off,
                                                             const int32_t* __restrict__ row_cnt, int32_t* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) float qlds[];  // dim floats (+ pad to 4) + kCosWaves partials
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = blockIdx.x / slices;
  const int slice = blockIdx.x % slices;
  int64_t obase = (int64_t)qi * c;
  if (RAGGED) {
    c = row_cnt[qi];
    obase = row_off[qi];
    if (slice * ((c + slices - 1) / slices) >= c) return;     // (nothing in this slice: the whole workgroup leaves together)
  }
  const int dim4 = (dim + 3) & ~3;
  const float* __restrict__ qv = queries + (int64_t)qi * dim;

  float qq = 0.f;
  for (int k = tid; k < dim4; k += kCosThreads) {
    const float v = k < dim ? qv[k] : 0.f;
    qlds[k] = v;
    qq = __builtin_fmaf(v, v, qq);
  }
  qq = wave_sum(qq);
  float* part = qlds + dim4;
  if (lane == 0) part[wave] = qq;
  __syncthreads();
  float qnorm2 = 0.f;
#pragma unroll
  for (int w = 0; w < kCosWaves; ++w) qnorm2 += part[w];
  const float qnorm = sqrtf(qnorm2);
  if (slice == 0 && tid == 0 && qstatus != nullptr) qstatus[qi] = (qnorm == 0.f) ? 1 : 0;
  if (RAGGED && slice == 0 && tid == 0 && err != nullptr && qnorm == 0.f) atomicOr(err, 4);

  // candidates of this slice, dealt to waves in groups of kCosInflight
  const int per_slice = (c + slices - 1) / slices;
  const int c_begin = slice * per_slice;
  const int c_end = min(c, c_begin + per_slice);

  for (int base = c_begin + wave * kCosInflight; base < c_end; base += kCosWaves * kCosInflight) {
    const float* rowp[kCosInflight];
    int st[kCosInflight];
#pragma unroll
    for (int u = 0; u < kCosInflight; ++u) {
      const int ci = base + u;
      int64_t idx = 0;
      st[u] = 3;  // 3 = not a candidate (past the end)
      if (ci < c_end) {
        idx = cand_idx != nullptr ? cand_idx[obase + ci] : obase + ci;
        st[u] = (idx < 0 || idx >= m) ? 2 : 0;
      }
      rowp[u] = corpus + (st[u] == 0 ? idx : 0) * ldc;
    }
    float dot[kCosInflight], nn[kCosInflight];
#pragma unroll
    for (int u = 0; u < kCosInflight; ++u) { dot[u] = 0.f; nn[u] = 0.f; }

    if (ALIGNED) {
      for (int k = lane * 4; k < dim; k += 256) {
        const f32x4 qx = *reinterpret_cast<const f32x4*>(qlds + k);
        f32x4 cx[kCosInflight];
#pragma unroll
        for (int u = 0; u < kCosInflight; ++u) cx[u] = *reinterpret_cast<const f32x4*>(rowp[u] + k);
#pragma unroll
        for (int u = 0; u < kCosInflight; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dot[u] = __builtin_fmaf(cx[u][e], qx[e], dot[u]);
            nn[u] = __builtin_fmaf(cx[u][e], cx[u][e], nn[u]);
          }
      }
    } else {
      for (int k = lane; k < dim; k += 64) {
        const float qx = qlds[k];
#pragma unroll
        for (int u = 0; u < kCosInflight; ++u) {
          const float cx = rowp[u][k];
          dot[u] = __builtin_fmaf(cx, qx, dot[u]);
          nn[u] = __builtin_fmaf(cx, cx, nn[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kCosInflight; ++u) {
      const float d = wave_sum(dot[u]);
      const float s2 = wave_sum(nn[u]);
      if (lane == 0 && st[u] != 3) {
        const int64_t o = obase + base + u;
        int code = st[u];
        float sc;
        if (code == 0) {
          const float cn = sqrtf(s2);
          if (cn == 0.f) code = 1;
          sc = d / (cn * qnorm);
        }
        if (code != 0) sc = __builtin_nanf("");
        scores[o] = sc;
        if (status != nullptr) status[o] = (uint8_t)code;
        if (RAGGED && code != 0 && err != nullptr) atomicOr(err, code);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// l2_norm: out = x / ||x||, one workgroup per row (reference helper lshrs/utils/norm.py:48-61)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ X, int64_t ldx, int dim,
                                                           float* __restrict__ out, uint8_t* __restrict__ status) {
  __shared__ float part[4];
  const int64_t row = blockIdx.x;
  const float* x = X + row * ldx;
  float ss = 0.f;
  for (int k = threadIdx.x; k < dim; k += 256) ss = __builtin_fmaf(x[k], x[k], ss);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float norm = sqrtf(part[0] + part[1] + part[2] + part[3]);
  if (threadIdx.x == 0 && status != nullptr) status[row] = (norm == 0.f) ? 1 : 0;
  float* o = out + row * (int64_t)dim;
  for (int k = threadIdx.x; k < dim; k += 256) o[k] = x[k] / norm;
}

// ------------------------------------------------------------------------------------------
// K3: descending order of each query's scores: bitonic network over 64-bit (key, position)
// pairs in LDS.  key ascending == score descending; NaN last; ties by ascending position.
// ------------------------------------------------------------------------------------------
constexpr int kTopkThreads = 256;

__device__ __forceinline__ uint32_t desc_key(float f) {
  if (f != f) return 0xFFFFFFFFu;  // NaN: after everything
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
  return ~u;                                       // descending, and never 0xFFFFFFFF for non-NaN? (-inf -> 0xFF800000 -> fine)
}

__global__ __launch_bounds__(kTopkThreads) void topk_kernel(const float* __restrict__ scores, int c, int cpad, int k,
                                                            int32_t* __restrict__ order, float* __restrict__ sorted) {
  extern __shared__ __attribute__((aligned(16))) uint64_t items[];
  const int qi = blockIdx.x;
  const float* s = scores + (int64_t)qi * c;
  for (int t = threadIdx.x; t < cpad; t += kTopkThreads) {
    uint64_t v = ~0ull;  // padding sorts after every real item (position field > any real position)
    if (t < c) v = ((uint64_t)desc_key(s[t]) << 32) | (uint32_t)t;
    items[t] = v;
  }
  __syncthreads();
  for (int size = 2; size <= cpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (cpad >> 1); t += kTopkThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t a = items[lo], b = items[hi];
        if ((a > b) == up) {
          items[lo] = b;
          items[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < k; t += kTopkThreads) {
    const uint32_t pos = (uint32_t)items[t];
    order[(int64_t)qi * k + t] = (int32_t)pos;
    sorted[(int64_t)qi * k + t] = s[pos];
  }
}

// ---- lists longer than one LDS network: the same bitonic network over a global u64 array ------------
constexpr int kTopkChunk = 4096;  // items per workgroup-local stage (32 KiB of LDS)

__global__ void topk_fill_kernel(const float* __restrict__ scores, int c, int64_t cpad, uint64_t* __restrict__ items) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpad) return;
  const int qi = blockIdx.y;
  uint64_t v = ~0ull;
  if (t < c) v = ((uint64_t)desc_key(scores[(int64_t)qi * c + t]) << 32) | (uint32_t)t;
  items[(int64_t)qi * cpad + t] = v;
}

// All compare-exchange steps with stride < kTopkChunk of one merge size (or, with full = true, the whole
// network up to size kTopkChunk) on a chunk held in LDS.  Direction follows the GLOBAL index.
__global__ __launch_bounds__(kTopkThreads) void topk_local_kernel(uint64_t* __restrict__ items, int64_t cpad, int64_t size,
                                                                  bool full) {
  __shared__ uint64_t buf[kTopkChunk];
  const int qi = blockIdx.y;
  const int64_t base = (int64_t)blockIdx.x * kTopkChunk;
  uint64_t* g = items + (int64_t)qi * cpad + base;
  for (int t = threadIdx.x; t < kTopkChunk; t += kTopkThreads) buf[t] = g[t];
  __syncthreads();
  const int64_t first = full ? 2 : size;
  const int64_t last = full ? kTopkChunk : size;
  for (int64_t sz = first; sz <= last; sz <<= 1) {
    const int top = (int)((sz < (int64_t)kTopkChunk ? sz : (int64_t)kTopkChunk) >> 1);
    for (int stride = top; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (kTopkChunk >> 1); t += kTopkThreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (((base + lo) & sz) == 0);
        const uint64_t a = buf[lo], b = buf[hi];
        if ((a > b) == up) {
          buf[lo] = b;
          buf[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < kTopkChunk; t += kTopkThreads) g[t] = buf[t];
}

__global__ void topk_global_step_kernel(uint64_t* __restrict__ items, int64_t cpad, int64_t size, int64_t stride) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (cpad >> 1)) return;
  uint64_t* g = items + (int64_t)blockIdx.y * cpad;
  const int64_t lo = 2 * t - (t & (stride - 1));
  const int64_t hi = lo + stride;
  const bool up = ((lo & size) == 0);
  const uint64_t a = g[lo], b = g[hi];
  if ((a > b) == up) {
    g[lo] = b;
    g[hi] = a;
  }
}

__global__ void topk_emit_kernel(const float* __restrict__ scores, const uint64_t* __restrict__ items, int c, int64_t cpad,
                                 int k, int32_t* __restrict__ order, float* __restrict__ sorted) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const int qi = blockIdx.y;
  const uint32_t pos = (uint32_t)items[(int64_t)qi * cpad + t];
  order[(int64_t)qi * k + t] = (int32_t)pos;
  sorted[(int64_t)qi * k + t] = scores[(int64_t)qi * c + pos];
}

inline int64_t topk_pad(int64_t c) {
  int64_t cpad = kTopkChunk;
  while (cpad < c) cpad <<= 1;
  return cpad;
}
}  // namespace

// Ascending sort of every row of a (q, cpad) array of 64-bit items in global memory (cpad a power of two >= kTopkChunk): the
// network K3 runs for long lists, for the candidate path's lists that do not fit LDS (query.hip).
int lshrs_sort_u64_rows(uint64_t* items, int q, int64_t cpad, hipStream_t s) {
  if (q <= 0 || q > 65535 || cpad < kTopkChunk || (cpad & (cpad - 1)) != 0) return LSHRS_E_BADARG;
  const dim3 half((unsigned)(((cpad >> 1) + 255) / 256), (unsigned)q);
  const dim3 chunks((unsigned)(cpad / kTopkChunk), (unsigned)q);
  hipLaunchKernelGGL(topk_local_kernel, chunks, dim3(kTopkThreads), 0, s, items, cpad, (int64_t)kTopkChunk, true);
  for (int64_t size = 2 * (int64_t)kTopkChunk; size <= cpad; size <<= 1) {
    for (int64_t stride = size >> 1; stride >= kTopkChunk; stride >>= 1)
      hipLaunchKernelGGL(topk_global_step_kernel, half, dim3(256), 0, s, items, cpad, size, stride);
    hipLaunchKernelGGL(topk_local_kernel, chunks, dim3(kTopkThreads), 0, s, items, cpad, size, false);
  }
  return -(int)hipGetLastError();
}

extern "C" {

int lshrs_cosine_batch_f32(const float* corpus, int64_t m, int64_t ldc, int32_t dim, const float* queries, int32_t q,
                           const int64_t* cand_idx, int32_t c, float* scores, uint8_t* status, uint8_t* qstatus,
                           void* stream) {
  if (q == 0 || c == 0) return 0;
  if (corpus == nullptr || queries == nullptr || scores == nullptr || m <= 0 || dim <= 0 || q < 0 || c < 0 || ldc < dim)
    return LSHRS_E_BADARG;
  if (dim > 16384) return LSHRS_E_TOOLARGE;
  if (cand_idx == nullptr && (int64_t)q * c > m) return LSHRS_E_BADARG;
  // enough workgroups to fill 256 CUs several times over even for a single query
  int slices = 1;
  const int per_block = kCosWaves * kCosInflight;
  while ((int64_t)q * slices < 4096 && (c + slices - 1) / slices > 2 * per_block) slices *= 2;
  if ((int64_t)q * slices > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  const bool aligned = (dim % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0);
  const size_t shmem = (size_t)(((dim + 3) & ~3) + kCosWaves) * sizeof(float);
  const dim3 grid((unsigned)((int64_t)q * slices)), block(kCosThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (aligned)
    hipLaunchKernelGGL((cosine_kernel<true, false>), grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_idx, c, slices,
                       scores, status, qstatus, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((cosine_kernel<false, false>), grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_idx, c, slices,
                       scores, status, qstatus, nullptr, nullptr, nullptr);
  return -(int)hipGetLastError();
}

int lshrs_cosine_ragged_f32(const float* corpus, int64_t m, int64_t ldc, int32_t dim, const float* queries, int32_t q,
                            const int64_t* cand_rows, const int64_t* row_off, const int32_t* row_cnt, int64_t total,
                            float* scores, int32_t* err, void* stream) {
  if (q == 0 || total == 0) return 0;
  if (corpus == nullptr || queries == nullptr || cand_rows == nullptr || row_off == nullptr || row_cnt == nullptr ||
      scores == nullptr || m <= 0 || dim <= 0 || q < 0 || total < 0 || ldc < dim)
    return LSHRS_E_BADARG;
  if (dim > 16384) return LSHRS_E_TOOLARGE;
  // slices per query from the AVERAGE list (the lists of one batch are alike: a bucket per band each): enough workgroups to
  // fill 256 CUs several times over even for a handful of queries
  int slices = 1;
  const int per_block = kCosWaves * kCosInflight;
  const int64_t avg = (total + q - 1) / q;
  while ((int64_t)q * slices < 4096 && (avg + slices - 1) / slices > 2 * per_block && slices < 1024) slices *= 2;
  if ((int64_t)q * slices > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  const bool aligned = (dim % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(corpus) & 15) == 0);
  const size_t shmem = (size_t)(((dim + 3) & ~3) + kCosWaves) * sizeof(float);
  const dim3 grid((unsigned)((int64_t)q * slices)), block(kCosThreads);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (aligned)
    hipLaunchKernelGGL((cosine_kernel<true, true>), grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_rows, 0, slices,
                       scores, nullptr, nullptr, row_off, row_cnt, err);
  else
    hipLaunchKernelGGL((cosine_kernel<false, true>), grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_rows, 0, slices,
                       scores, nullptr, nullptr, row_off, row_cnt, err);
  return -(int)hipGetLastError();
}

int lshrs_l2_normalize_f32(const float* X, int64_t n, int64_t ldx, int32_t dim, float* out, uint8_t* status,
                           void* stream) {
  if (n == 0) return 0;
  if (X == nullptr || out == nullptr || n < 0 || dim <= 0 || ldx < dim) return LSHRS_E_BADARG;
  if (n > 0x7fffffffLL) return LSHRS_E_TOOLARGE;
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), X, ldx,
                     dim, out, status);
  return -(int)hipGetLastError();
}

int64_t lshrs_topk_workspace_bytes(int32_t q, int32_t c) {
  if (q < 0 || c < 0) return LSHRS_E_BADARG;
  if (c <= 16384) return 0;
  return (int64_t)q * topk_pad(c) * (int64_t)sizeof(uint64_t);
}

int lshrs_topk_desc_f32(const float* scores, int32_t q, int32_t c, int32_t k, int32_t* order, float* sorted,
                        void* workspace, void* stream) {
  if (q == 0 || k == 0) return 0;
  if (scores == nullptr || order == nullptr || sorted == nullptr || q < 0 || c <= 0 || k < 0 || k > c)
    return LSHRS_E_BADARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c <= 16384) {  // one LDS-resident network per query
    int cpad = 2;
    while (cpad < c) cpad <<= 1;
    const size_t shmem = (size_t)cpad * sizeof(uint64_t);
    if (shmem > 48 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (e != hipSuccess) return -(int)e;
    }
    hipLaunchKernelGGL(topk_kernel, dim3((unsigned)q), dim3(kTopkThreads), shmem, s, scores, c, cpad, k, order, sorted);
    return -(int)hipGetLastError();
  }
  if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 7)) return LSHRS_E_BADARG;
  if (q > 65535) return LSHRS_E_TOOLARGE;
  uint64_t* items = static_cast<uint64_t*>(workspace);
  const int64_t cpad = topk_pad(c);
  const dim3 qgrid((unsigned)((cpad + 255) / 256), (unsigned)q), half((unsigned)(((cpad >> 1) + 255) / 256), (unsigned)q);
  const dim3 chunks((unsigned)(cpad / kTopkChunk), (unsigned)q);
  hipLaunchKernelGGL(topk_fill_kernel, qgrid, dim3(256), 0, s, scores, c, cpad, items);
  hipLaunchKernelGGL(topk_local_kernel, chunks, dim3(kTopkThreads), 0, s, items, cpad, (int64_t)kTopkChunk, true);
  for (int64_t size = 2 * (int64_t)kTopkChunk; size <= cpad; size <<= 1) {
    for (int64_t stride = size >> 1; stride >= kTopkChunk; stride >>= 1)
      hipLaunchKernelGGL(topk_global_step_kernel, half, dim3(256), 0, s, items, cpad, size, stride);
    hipLaunchKernelGGL(topk_local_kernel, chunks, dim3(kTopkThreads), 0, s, items, cpad, size, false);
  }
  hipLaunchKernelGGL(topk_emit_kernel, dim3((unsigned)((k + 255) / 256), (unsigned)q), dim3(256), 0, s, scores, items, c,
                     cpad, k, order, sorted);
  return -(int)hipGetLastError();
}

}  // extern "C"
