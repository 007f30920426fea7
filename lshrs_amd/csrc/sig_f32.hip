// sig_f32.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// K1f: the exact-f32 signature pass on v_mfma_f32_32x32x2_f32 (ballot bit-pack), its diagnostic projection form.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {

// ---- ring-buffered main loop of the f32 kernel ----------------------------------------------------
// A 32-deep k-tile is handled as two HALVES (fragments q = 0,1 then q = 2,3 of every column tile: MFMA steps
// s = 0..7 and 8..15).  Halves go through a ring of three LDS buffers, staged two halves ahead, and the fragments of
// the next group are read while the current group's MFMAs issue, so no ds_read latency is exposed behind the
// barrier that ends each half.
template <int NT>
__device__ __forceinline__ void stage_p_half(const float* __restrict__ tile, int part, float* lds_buf, int tid) {
  constexpr int kBlocks = NT * 2;  // (jt, qq) fragment blocks of 1 KiB in one half
  const int wave = tid >> 6;
  const int lane = tid & 63;
#pragma unroll
  for (int base = 0; base < kBlocks; base += kSigWaves) {
    const int blk = base + wave;  // wave-uniform
    if (kBlocks % kSigWaves == 0 || blk < kBlocks) {
      const int jt = blk >> 1, qq = blk & 1;
      const float* g = tile + (size_t)(((jt * 4 + 2 * part + qq) * 64) + lane) * 4;
      float* l = lds_buf + (size_t)blk * kFragFloats;  // wave-uniform base; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)l, 16, 0, 0);
    }
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void load_x_half(const float* __restrict__ xrow, int kbase, int dim, f32x4 (&a)[2]) {
  // lane (i, h) owns k = kbase + 4qq + r (kbase already holds 32*kt + 16*h + 8*part): 32 contiguous bytes
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const int k = kbase + 4 * qq;
    if (ALIGNED) {
      if (k < dim)
        a[qq] = *reinterpret_cast<const f32x4*>(xrow + k);
      else
        a[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (k + r < dim) ? xrow[k + r] : 0.f;
      a[qq] = v;
    }
  }
}

template <int NT>
__device__ __forceinline__ void read_frags(const float* lds_buf, int qq, int lane, f32x4 (&b)[NT]) {
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
    b[jt] = *reinterpret_cast<const f32x4*>(lds_buf + ((jt * 2 + qq) * 64 + lane) * 4);
}

// One fragment group: 4 k-steps x NT column tiles of MFMAs.  Every accumulator tile sees its k-steps in the same
// order whatever NT is (the order oracle/chain_model.c restates).
template <int NT>
__device__ __forceinline__ void mfma_group(const f32x4& a, const f32x4 (&b)[NT], f32x16 (&acc)[NT], float& ss,
                                           float& amax) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float av = a[r];
    ss = __builtin_fmaf(av, av, ss);
    amax = __builtin_fmaxf(amax, __builtin_fabsf(av));
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[jt][r], acc[jt], 0, 0, 0);
  }
}

// Workgroup = 4 waves (one per SIMD, two workgroups per CU: the two waves sharing a SIMD belong to different
// workgroups and never wait at the same barrier); each wave owns one 32-row tile x all 32*NT columns of its column
// block (128 accumulator registers at NT = 8).  Variants measured and dropped in round 1 (8-wave workgroups, two row
// tiles per wave, whole-tile double buffering): profiles/r01_kernel_variants_ab.log.
template <int NT, bool ALIGNED, int MODE>
__global__ __launch_bounds__(kSigWaves * 64, 2) void sig_kernel(const SigArgs args) {
  constexpr bool PROJECT = MODE == 2;
  constexpr int kTileFloats = NT * 4 * kFragFloats;
  constexpr int kHalfFloats = NT * 2 * kFragFloats;
  constexpr int kStageFloats = 3 * kHalfFloats;          // ring of three halves
  constexpr int kBlockRows = kSigWaves * kRowsPerWave;
  __shared__ __attribute__((aligned(16))) float lds[kStageFloats + kBlockRows];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5;
  const int i = lane & 31;
  const int cb = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.x * kBlockRows + wave * kRowsPerWave;
  const int dim = args.dim;
  const int ktiles = args.ktiles;
  const float* __restrict__ img = args.image + (size_t)cb * ktiles * kTileFloats;
  const int64_t xr_ = row0 + i;
  const float* __restrict__ xrow = args.X + (xr_ < args.n ? xr_ : args.n - 1) * args.ldx;  // clamp: loads stay in bounds, stores are masked

  f32x16 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jt][r] = 0.f;
  float ss = 0.f, amax = 0.f;  // sum of squares / max |x| of this lane's share of its row

  unsigned long long t_shader = 0, t_real = 0;
  if (args.clock_probe != nullptr) {
    t_shader = __builtin_amdgcn_s_memtime();
    t_real = __builtin_amdgcn_s_memrealtime();
  }

  {
    const int halves = 2 * ktiles;
    f32x4 a_cur[2], a_nxt[2];
    f32x4 b0[NT], b1[NT];
    stage_p_half<NT>(img, 0, lds, tid);
    stage_p_half<NT>(img, 1, lds + kHalfFloats, tid);
    load_x_half<ALIGNED>(xrow, 16 * h, dim, a_cur);
    __syncthreads();
    read_frags<NT>(lds, 0, lane, b0);
    // land b0 before the loop, so that on every path into the loop header nothing is pending and the
    // compiler's wait before group 1 can be a counted lgkmcnt (b1 only), not a drain
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
    for (int hh = 0; hh < halves; ++hh) {
      const float* cur = lds + (hh % 3) * kHalfFloats;
      if (hh + 2 < halves)
        stage_p_half<NT>(img + (size_t)((hh + 2) >> 1) * kTileFloats, (hh + 2) & 1, lds + ((hh + 2) % 3) * kHalfFloats, tid);
      if (hh + 1 < halves)
        load_x_half<ALIGNED>(xrow, ((hh + 1) >> 1) * kKTile + 16 * h + 8 * ((hh + 1) & 1), dim, a_nxt);
      // Issue order is pinned (sched_barrier): left alone, the scheduler sinks each ds_read group down to its
      // first use and the wave then sits out the LDS latency with the matrix pipe idle.
      read_frags<NT>(cur, 1, lane, b1);                       // lands while group 0 issues
      __builtin_amdgcn_sched_barrier(0);
      mfma_group<NT>(a_cur[0], b0, acc, ss, amax);
      // b1 was issued a whole group (32 MFMAs) ago: this wait is free, and taking it BEFORE the next reads are
      // issued keeps it from turning into a drain of those reads (hipcc emits lgkmcnt(0), not a counted wait)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      // next half's first fragments (visible since the last barrier).  Unconditional on purpose: after the last
      // half this reads a stale ring slot that nobody uses, which keeps the wait counters branch-free.
      read_frags<NT>(lds + ((hh + 1) % 3) * kHalfFloats, 0, lane, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group<NT>(a_cur[1], b1, acc, ss, amax);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      a_cur[0] = a_nxt[0];
      a_cur[1] = a_nxt[1];
    }
  }

  if (args.clock_probe != nullptr && tid == 0) {
    const unsigned long long slot = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
    args.clock_probe[2 * slot] = __builtin_amdgcn_s_memtime() - t_shader;
    args.clock_probe[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - t_real;
  }

  // accumulator map (32x32 tile): column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (PROJECT) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < args.n) args.Y[row * args.ldy + (cb * NT + jt) * 32 + i] = acc[jt][r];
      }
    return;
  }

  // ---- row statistics: ||x||, zero-vector flag ------------------------------------------
  float* norm_lds = lds + kStageFloats + wave * kRowsPerWave;
  {
    const float s2 = ss + __shfl_xor(ss, 32);
    const float am = __builtin_fmaxf(amax, __shfl_xor(amax, 32));
    const int64_t myrow = row0 + i;
    if (h == 0) {
      norm_lds[i] = sqrtf(s2) * args.tau;
      if (cb == 0 && args.row_flags != nullptr && myrow < args.n) {
        const bool has_nan = s2 != s2;
        const bool zero = (am <= 1e-8f) && !has_nan;
        args.row_flags[myrow] = (uint8_t)((zero ? 1 : 0) | (has_nan ? 2 : 0));
      }
    }
  }
  __syncthreads();

  constexpr int LPR = NT >= 2 ? 2 : 1;   // lanes that hold one output row
  constexpr int WPL = NT >= 2 ? NT / 2 : 1;  // 32-bit words per lane
  constexpr bool want_ties = MODE == 1;

  // ---- sign bits + tie bits of the 32-row tile, one ballot per accumulator register --------------
  uint32_t kw[WPL], tw[WPL];
#pragma unroll
  for (int w = 0; w < WPL; ++w) { kw[w] = 0u; tw[w] = 0u; }
  f32x4 rn[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rn[g] = *reinterpret_cast<const f32x4*>(norm_lds + 8 * g + 4 * h);
  // wave-uniform screen for ties: |y| < (largest tau*||x|| of the tile's rows) * (largest ||p|| of the block)
  float screen = 0.f;
  if (want_ties) {
    float m = __builtin_fmaxf(__builtin_fmaxf(rn[0][0], rn[0][1]), __builtin_fmaxf(rn[0][2], rn[0][3]));
#pragma unroll
    for (int g = 1; g < 4; ++g)
      m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(rn[g][0], rn[g][1]), __builtin_fmaxf(rn[g][2], rn[g][3])));
    m = __builtin_fmaxf(m, __shfl_xor(m, 32));  // lanes of one half hold 16 of the 32 rows
    // NaN norms (a NaN in x) must not hide the finite rows next to them: fmaxf drops NaNs, so m is the
    // largest finite norm; rows that are NaN produce NaN projections, which never tie.
    screen = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m))) * args.norm_max[cb];
  }

#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float y = acc[jt][r];
      const int rho = (r & 3) + 8 * (r >> 2);
      const int l0 = rho * LPR + jt / WPL;         // lane receiving the word of row rho
      const int l1 = (rho + 4) * LPR + jt / WPL;   // lane receiving the word of row rho + 4
      deposit_positive(kw[jt % WPL], y, l0, l1);   // bit = (y > 0): 0, -0 and NaN give 0 (lsh.py:204)
      if (want_ties) any |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(y) < screen);
    }
    if (any != 0) {  // wave-uniform, rare (a few % of column tiles): the exact per-element test
      const float pn = args.norms[(cb * NT + jt) * 32 + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float thr = rn[r >> 2][r & 3] * pn;
        // strict '<': thr == 0 (zero x, zero-padded column) never ties
        const int rho = (r & 3) + 8 * (r >> 2);
        deposit_abs_below(tw[jt % WPL], acc[jt][r], thr, rho * LPR + jt / WPL, (rho + 4) * LPR + jt / WPL);
      }
    }
  }

  // ---- stores: lane L holds words [ (L % LPR) * WPL, +WPL ) of row L / LPR -----------------
  const int orow = lane / LPR;
  const int64_t grow = row0 + orow;
  const bool lane_on = (NT >= 2 || lane < 32) && grow < args.n;
  const int word0 = cb * NT + (lane % LPR) * WPL;  // first 32-column word this lane holds
  const int byte0 = word0 * 4;
  if (lane_on) {
    uint8_t* dst = args.keys + grow * (int64_t)args.row_bytes + byte0;
    if (args.vec_store && byte0 + 4 * WPL <= args.row_bytes) {
      if (WPL == 4) {
        *reinterpret_cast<u32x4*>(dst) = u32x4{kw[0], kw[1 % WPL], kw[2 % WPL], kw[3 % WPL]};
      } else if (WPL == 2) {
        *reinterpret_cast<u32x2*>(dst) = u32x2{kw[0], kw[1 % WPL]};
      } else {
        *reinterpret_cast<uint32_t*>(dst) = kw[0];
      }
    } else {
#pragma unroll
      for (int w = 0; w < WPL; ++w)
#pragma unroll
        for (int bsel = 0; bsel < 4; ++bsel)
          if (byte0 + 4 * w + bsel < args.row_bytes) dst[4 * w + bsel] = (uint8_t)(kw[w] >> (8 * bsel));
    }
    if (want_ties) {
#pragma unroll
      for (int w = 0; w < WPL; ++w) {
        if (tw[w] != 0u) {
          const int slot = atomicAdd(args.tie_count, 1);
          if (slot < args.tie_cap) {
            args.tie_list[2 * (int64_t)slot] = (grow + args.row_base) * 65536 + (word0 + w);
            args.tie_list[2 * (int64_t)slot + 1] = (int64_t)tw[w];
          }
        }
      }
    }
  }
}

template <int NT>
int launch_sig(const SigArgs& a, const SigGeom& g, bool aligned, bool project, hipStream_t s) {
  const int mode = project ? 2 : (a.tie_list != nullptr ? 1 : 0);
  constexpr int block_rows = kSigWaves * kRowsPerWave;
  const dim3 grid((unsigned)((a.n + block_rows - 1) / block_rows), (unsigned)g.cb, 1);
  const dim3 block(kSigWaves * 64, 1, 1);
#define LSHRS_LAUNCH(AL, MD) hipLaunchKernelGGL((sig_kernel<NT, AL, MD>), grid, block, 0, s, a)
  if (aligned) {
    if (mode == 0) LSHRS_LAUNCH(true, 0);
    else if (mode == 1) LSHRS_LAUNCH(true, 1);
    else LSHRS_LAUNCH(true, 2);
  } else {
    if (mode == 0) LSHRS_LAUNCH(false, 0);
    else if (mode == 1) LSHRS_LAUNCH(false, 1);
    else LSHRS_LAUNCH(false, 2);
  }
#undef LSHRS_LAUNCH
  return -(int)hipGetLastError();
}

int dispatch_sig(const SigArgs& a, const SigGeom& g, bool project, hipStream_t s) {
  const bool aligned = (a.dim % 4 == 0) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.X) & 15) == 0);
  switch (g.nt) {
    case 8: return launch_sig<8>(a, g, aligned, project, s);
    case 4: return launch_sig<4>(a, g, aligned, project, s);
    case 2: return launch_sig<2>(a, g, aligned, project, s);
    default: return launch_sig<1>(a, g, aligned, project, s);
  }
}

// ------------------------------------------------------------------------------------------
// tie-break plumbing
}  // namespace

extern "C" {

int lshrs_sig_hash_batch_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                             int32_t rows_per_band, int32_t dim, uint8_t* keys, int64_t* tie_list, int32_t tie_cap,
                             int32_t* tie_count, float tau, uint8_t* row_flags, const lshrs_sig_opts* opts, void* stream) {
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || keys == nullptr || n < 0 || ldx < dim ||
      !sig_shape_ok(num_bands, rows_per_band, dim))
    return LSHRS_E_BADARG;
  if (tie_list != nullptr && (tie_count == nullptr || tie_cap < 0)) return LSHRS_E_BADARG;
  if (n >= ((int64_t)1 << 47)) return LSHRS_E_TOOLARGE;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  if ((n + 127) / 128 > 0x7fffffffLL || g.cb > 65535) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* base = static_cast<const float*>(workspace);
  const int row_bytes = num_bands * g.bb;
  // One launch covers rows [lo, hi) with either geometry.
  auto launch = [&](int64_t lo, int64_t hi, bool fine) -> int {
    const SigGeom gg = fine ? sig_fine_geom(g) : g;
    SigArgs a{};
    a.X = X + lo * ldx;
    a.n = hi - lo;
    a.ldx = ldx;
    a.dim = dim;
    a.ktiles = gg.ktiles;
    a.norms = base + sig_image_floats(g);                       // per padded column: shared by both geometries
    if (fine) {
      a.image = base + sig_main_floats(g);
      a.norm_max = a.image + sig_image_floats(gg);
    } else {
      a.image = base;
      a.norm_max = a.norms + sig_norm_floats(g);
    }
    if (!(tau > 0.f)) {      // LSHRS_WINDOW_PROVEN: the tie window of lshrs_sig_set_window (coefficient per column, factor 1)
      const SigWindow w = sig_window(base, g);
      a.norms = w.wt;
      a.norm_max = fine ? w.wtmax_fine : w.wtmax;
    }
    a.keys = keys + lo * row_bytes;
    a.row_bytes = row_bytes;
    const int wpl_bytes = gg.nt >= 2 ? 2 * gg.nt : 4;  // bytes one lane stores
    a.vec_store = (row_bytes % wpl_bytes == 0) && ((reinterpret_cast<uintptr_t>(a.keys) % wpl_bytes) == 0);
    a.row_base = lo;
    a.tie_list = tie_list;
    a.tie_cap = tie_cap;
    a.tie_count = tie_count;
    a.tau = tau > 0.f ? tau : 1.0f;
    a.row_flags = row_flags != nullptr ? row_flags + lo : nullptr;
    a.clock_probe = (lo == 0 && !fine) ? read_opts(opts).clock_probe : nullptr;
    return dispatch_sig(a, gg, false, s);
  };
  // Whole rounds of NT-wide workgroups first; what is left (less than one round) takes the fine geometry when
  // that finishes sooner than one more full-length, mostly idle round.
  const bool fine_ok = sig_has_fine(g) && g.tiles32 <= 65535;
  const int64_t n_main = (n / kRoundRows) * kRoundRows;
  const int64_t tail = n - n_main;
  if (n_main > 0) {
    const int rc = launch(0, n_main, false);
    if (rc != 0) return rc;
  }
  if (tail > 0) return launch(n_main, n, fine_ok && sig_prefer_fine(g, tail));
  return 0;
}

int lshrs_sig_project_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                          int32_t rows_per_band, int32_t dim, float* Y, int64_t ldy, void* stream) {
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || Y == nullptr || n < 0 || ldx < dim ||
      !sig_shape_ok(num_bands, rows_per_band, dim))
    return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  if (ldy < (int64_t)g.cb * g.nt * 32) return LSHRS_E_BADARG;
  if ((n + 127) / 128 > 0x7fffffffLL || g.cb > 65535) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* base = static_cast<const float*>(workspace);
  auto launch = [&](int64_t lo, int64_t hi, bool fine) -> int {  // same launch plan as the hashing entry point
    const SigGeom gg = fine ? sig_fine_geom(g) : g;
    SigArgs a{};
    a.X = X + lo * ldx;
    a.n = hi - lo;
    a.ldx = ldx;
    a.dim = dim;
    a.ktiles = gg.ktiles;
    a.norms = base + sig_image_floats(g);
    a.image = fine ? base + sig_main_floats(g) : base;
    a.norm_max = fine ? a.image + sig_image_floats(gg) : a.norms + sig_norm_floats(g);
    a.Y = Y + lo * ldy;
    a.ldy = ldy;
    return dispatch_sig(a, gg, true, s);
  };
  const bool fine_ok = sig_has_fine(g) && g.tiles32 <= 65535;
  const int64_t n_main = (n / kRoundRows) * kRoundRows;
  if (n_main > 0) {
    const int rc = launch(0, n_main, false);
    if (rc != 0) return rc;
  }
  if (n > n_main) return launch(n_main, n, fine_ok && sig_prefer_fine(g, n - n_main));
  return 0;
}

}  // extern "C"
