// sig_small.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// K1s: a query vector or a handful - every projection the replayed host-BLAS value, one memory round trip.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
// ------------------------------------------------------------------------------------------
// A query vector or a handful of them (LSHRS.ingest / get_top_k / hash_vector: the reference's own calling pattern).
// No first pass at all: EVERY projection is the replayed host-BLAS value (blas_model 1, see sig_fix8_kernel), so the
// keys are the reference's by construction and the kernel is one memory round trip deep instead of the f32 kernel's
// ktiles dependent stages: one single-wave workgroup per KEY BYTE = eight projections of one row.  Everything the wave
// needs is in flight at once - the eight hyperplanes (LDS-DMA, one instruction per k-tile, [k-tile][chunk][g] as in
// stage 2) and the x row once (the eight projections share it: 64 chunks per instruction) - then 4 fmas per k-tile and
// lane, the library's reduction by three shuffles, a ballot for the byte.  x may live in pinned host memory (a few rows:
// the row crosses PCIe once per key byte, 32 x 3 KB per 768-d vector) or on the device.
// Completion without a stream wait: the last wave of every row takes a ticket, the last row publishes the tie count and
// the caller's epoch to host-visible memory after a system-scope fence: the host polls that word (no copy, no
// hipStreamSynchronize on the single-vector path).  Tickets are per row first (64 ints apart: one address sees at most
// row_bytes atomics) because returning atomics on ONE address serialise at ~40 ns each (export_counts_kernel's comment).
// ------------------------------------------------------------------------------------------
struct SmallArgs {
  const float* X;
  int64_t ldx;
  int ktiles;
  const float* prow;
  const float* norms;
  uint8_t* keys;          // device or host-visible
  int row_bytes;
  uint8_t* row_flags;     // optional
  int* counters;          // device, zeroed once by the caller: [0] ties, [1] rows done, [64 (1 + row)] key bytes of the row done
  int* host_done;         // optional, host-visible: [0] ties, [1] epoch
  int epoch;
  int n;
  float tau;
  int rows_per_band;      // (which of the library's kernels computes a column: blas_row_kind)
  int band_cols;
  int dim;                // (GENERAL: rows that are not whole k-tiles - X is readable, not used, up to 32 * ktiles per row)
};

template <int KT, bool GENERAL>
__global__ __launch_bounds__(64) void sig_small_kernel(const SmallArgs a) {
  static_assert(KT % 8 == 0, "the x row lands in whole 64-chunk instructions");
  __shared__ __attribute__((aligned(16))) f32x4 ps[KT * 64];
  __shared__ __attribute__((aligned(16))) f32x4 xs[KT * 8];
  const int lane = threadIdx.x, g = lane & 7, sub = lane >> 3;
  const int row = blockIdx.x / a.row_bytes, byte = blockIdx.x % a.row_bytes;
  const int col = 8 * byte + g;
  const size_t ldp = (size_t)a.ktiles * kKTile;
  // GENERAL: a row of 8 m + 4 elements gives its first four to the low lanes before the tiles begin at the fifth, and the
  // elements past its end read as zero (sig_fix8_kernel's comment); the x row sits in LDS from element 0 either way
  const int head = GENERAL ? (a.dim & 4) : 0, hq = head >> 2;
  const int body = GENERAL ? a.dim - head : a.ktiles * kKTile;
  const int kt = GENERAL ? (body + kKTile - 1) / kKTile : a.ktiles;
  const float* pg = a.prow + (size_t)col * ldp + head + 4 * sub;
  const float* xg = a.X + (int64_t)row * a.ldx;
  const int xchunks = a.ktiles * 8;
  for (int b = 0; b * 64 < xchunks; ++b) {          // lanes past the row's end re-fetch its last chunk (lands unused)
    const int c = b * 64 + lane < xchunks ? b * 64 + lane : xchunks - 1;
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(xg + 4 * c), (LDS_AS void*)(xs + b * 64), 16, 0, 0);
  }
  for (int t = 0; t < kt; ++t)
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(pg + (size_t)t * kKTile), (LDS_AS void*)(ps + t * 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float* xf = reinterpret_cast<const float*>(xs);
  const float* pf = reinterpret_cast<const float*>(ps);
  float pj = 0.f, ss = 0.f, am = 0.f;
  const int kind = GENERAL ? blas_row_kind(col % a.band_cols, a.rows_per_band) : 0;
  if (GENERAL && head != 0) {
    const int hl = kind == 1 ? (sub & 3) : sub;
    if (hl < 4) {
      const float hx = xf[hl];
      pj = a.prow[(size_t)col * ldp + hl] * hx;
      if (sub < 4) {
        ss = hx * hx;
        am = __builtin_fabsf(hx);
      }
    }
  }
  for (int t = 0; t < kt; ++t) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {                   // k = head + 32 t + 8 m + sub: chunk 2 m + (sub >> 2), element sub & 3
      const int o = t * 8 + 2 * m + (sub >> 2);
      const int ox = GENERAL ? (o + hq < xchunks ? o + hq : xchunks - 1) : o;
      const bool in = !GENERAL || t * kKTile + 8 * m + sub < body;       // (past the row's end both factors read as zero)
      const float xv = in ? xf[ox * 4 + (sub & 3)] : 0.f;
      const float pv = in ? pf[(o * kFixG + g) * 4 + (sub & 3)] : 0.f;
      if (GENERAL && kind == 1) {                   // chain l = sub & 3: k = 8 m + l, then k = 8 m + 4 + l
        const int o0 = t * 8 + 2 * m, o1 = o0 + 1;
        const int kl = t * kKTile + 8 * m + (sub & 3);
        const int x0 = o0 + hq < xchunks ? o0 + hq : xchunks - 1, x1 = o1 + hq < xchunks ? o1 + hq : xchunks - 1;
        pj = mul_then_add(pj, kl < body ? pf[(o0 * kFixG + g) * 4 + (sub & 3)] : 0.f, kl < body ? xf[x0 * 4 + (sub & 3)] : 0.f);
        pj = mul_then_add(pj, kl + 4 < body ? pf[(o1 * kFixG + g) * 4 + (sub & 3)] : 0.f,
                          kl + 4 < body ? xf[x1 * 4 + (sub & 3)] : 0.f);
      } else if (GENERAL && kind == 2) {
        pj = mul_then_add(pj, pv, xv);
      } else {
        pj = __builtin_fmaf(pv, xv, pj);
      }
      ss = __builtin_fmaf(xv, xv, ss);
      am = __builtin_fmaxf(am, __builtin_fabsf(xv));
    }
  }
  const float yb = blas_reduce(pj, kind, lane);                    // the library's reduction: see sig_fix8_kernel
  float s2 = ss + __shfl(ss, (lane + 32) & 63);
  s2 += __shfl(s2, (lane + 8) & 63);
  s2 += __shfl(s2, (lane + 16) & 63);
  float m2 = __builtin_fmaxf(am, __shfl(am, (lane + 32) & 63));
  m2 = __builtin_fmaxf(m2, __shfl(m2, (lane + 8) & 63));
  m2 = __builtin_fmaxf(m2, __shfl(m2, (lane + 16) & 63));
  const bool want = sub == 0 && yb > 0.f;                          // (0, -0 and NaN give 0: lsh.py:204)
  const bool tie = sub == 0 && __builtin_fabsf(yb) < a.tau * sqrtf(s2) * a.norms[col];
  const unsigned bits = (unsigned)(__ballot(want) & 0xffull);      // lane g = bit g of the byte
  const int nt = __popcll(__ballot(tie) & 0xffull);
  if (lane == 0) {
    a.keys[(int64_t)row * a.row_bytes + byte] = (uint8_t)bits;
    if (byte == 0 && a.row_flags != nullptr) {
      const bool has_nan = s2 != s2;
      const bool zero = (m2 <= 1e-8f) && !has_nan;
      a.row_flags[row] = (uint8_t)((zero ? 1 : 0) | (has_nan ? 2 : 0));
    }
    if (nt != 0) atomicAdd(a.counters, nt);
    if (a.host_done != nullptr) {
      __threadfence_system();                                      // this wave's bytes are out before its ticket
      int* row_ticket = a.counters + 64 * (1 + row);
      if (atomicAdd(row_ticket, 1) == a.row_bytes - 1) {
        atomicExch(row_ticket, 0);
        if (atomicAdd(a.counters + 1, 1) == a.n - 1) {
          atomicExch(a.counters + 1, 0);
          const int ties = atomicExch(a.counters, 0);
          __atomic_store_n(a.host_done, ties, __ATOMIC_RELAXED);
          __threadfence_system();
          __atomic_store_n(a.host_done + 1, a.epoch, __ATOMIC_RELAXED);
        }
      }
    }
  }
}
}  // namespace

extern "C" {

int lshrs_sig_hash_small_replay_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                                    int32_t rows_per_band, int32_t dim, uint8_t* keys, uint8_t* row_flags,
                                    int32_t* counters, float tau, int32_t blas_model, int32_t* host_done, int32_t epoch,
                                    void* stream) {
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || keys == nullptr || counters == nullptr || n < 0 || ldx < dim ||
      !sig_shape_ok(num_bands, rows_per_band, dim) || blas_model != 1)
    return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  const int row_bytes = num_bands * g.bb;
  // (rows that are not whole k-tiles: the kernel fetches 32 * ktiles floats of every row - the caller pads the rows - and
  //  uses `dim` of them)
  if (dim % 4 != 0 || dim < 8 || ldx < (int64_t)g.ktiles * kKTile || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 ||
      g.ktiles > 128 || n > LSHRS_SMALL_MAX_ROWS || n * row_bytes > 0x7fffffffLL)
    return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* base = static_cast<const float*>(workspace);
  SmallArgs a{};
  a.X = X;
  a.ldx = ldx;
  a.ktiles = g.ktiles;
  a.prow = base + sig_rowmajor_offset_floats(g);
  a.norms = base + sig_image_floats(g);
  a.keys = keys;
  a.row_bytes = row_bytes;
  a.row_flags = row_flags;
  a.counters = counters;
  a.host_done = host_done;
  a.epoch = epoch;
  a.n = (int)n;
  a.tau = tau;
  a.rows_per_band = rows_per_band;
  a.band_cols = 8 * g.bb;
  a.dim = dim;
  const dim3 grid((unsigned)(n * row_bytes)), block(64);
  if (!blas_general(rows_per_band, g.ktiles, dim)) {
    if (g.ktiles <= 24) hipLaunchKernelGGL((sig_small_kernel<24, false>), grid, block, 0, s, a);
    else if (g.ktiles <= 48) hipLaunchKernelGGL((sig_small_kernel<48, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((sig_small_kernel<128, false>), grid, block, 0, s, a);      // 144 KiB of LDS: one workgroup per CU
  } else {
    if (g.ktiles <= 24) hipLaunchKernelGGL((sig_small_kernel<24, true>), grid, block, 0, s, a);
    else if (g.ktiles <= 48) hipLaunchKernelGGL((sig_small_kernel<48, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((sig_small_kernel<128, true>), grid, block, 0, s, a);
  }
  return -(int)hipGetLastError();
}

}  // extern "C"
