// sig_split.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// The split-precision signature pass as the C ABI offers it: stage 1 (sig16.hip / sig16r.hip) + stage 2 (sig_replay.hip) on
// one stream, with the tie replay; the library's version and build flags.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

#include <chrono>
#include <cstdlib>

using namespace lshrs;

extern "C" {

int lshrs_abi_version(void) { return LSHRS_ABI_VERSION; }

// Which measurement switches this build was compiled with (include/lshrs_hip.h, LSHRS_BUILD_*): every translation unit
// reports its own; the product build returns 0.
uint32_t lshrs_build_flags(void) { return lshrs_flags_sig16() | lshrs_flags_sig16r() | lshrs_flags_replay() | lshrs_flags_query(); }

int lshrs_stream_synchronize(void* stream) { return -(int)hipStreamSynchronize(static_cast<hipStream_t>(stream)); }

int lshrs_wait_done(const int32_t* done_host, int32_t epoch, int64_t spin_ns, void* stream) {
  if (done_host != nullptr && spin_ns > 0) {
    const volatile int32_t* word = done_host;
    const auto t0 = std::chrono::steady_clock::now();
    for (int64_t spins = 0;; ++spins) {
      if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == epoch) return 0;
#if !defined(__HIP_DEVICE_COMPILE__)
      __builtin_ia32_pause();
#endif
      if ((spins & 255) == 255 &&
          std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() > spin_ns)
        break;
    }
  }
  return -(int)hipStreamSynchronize(static_cast<hipStream_t>(stream));
}

// blas_model 0: ties are reported in tie_list (the caller resolves them on the host); > 0: stage 2 resolves them itself
// by replaying that summation order of the host BLAS (sig_fix8_kernel<true>), tie_list is not used.
// counters (replay only): the LSHRS_SIG_DEVICE_COUNTERS block; flag_y: the stage-1 value of every list entry (may be NULL).
// Which workgroup of sig16_kernel a launch takes: 128 rows, two per CU (sig16.hip), for vectors of up to kHalfMaxTiles k-tiles
// (measured, 1 M rows of 16 x 16: stage 1 +2 .. +5 % from 5 to 11 k-tiles, -1 .. -3 % from 12 on) and for batches whose
// 256-row workgroups would leave half of the CUs without one (up to 32 768 rows per column block: stage 1 59 -> 47 us at 768-d).
// LSHRS_SIG16_HALF_MAX_TILES (environment, read once; a measurement switch: tools/half_rows_ab.py) replaces both rules by a
// k-tile limit: 0 = never, 64 = always.
static bool sig16_half_rows(int ktiles, int64_t groups_of_256) {
  static const int forced = [] {
    const char* e = std::getenv("LSHRS_SIG16_HALF_MAX_TILES");
    return e != nullptr ? std::atoi(e) : -1;
  }();
  if (forced >= 0) return ktiles <= forced;
  return ktiles <= kHalfMaxTiles || groups_of_256 <= kHalfMaxGroups;
}

static int split_pass(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                      int32_t rows_per_band, int32_t dim, uint8_t* keys, int64_t* tie_list, int32_t tie_cap,
                      int32_t* tie_count, float tau, uint8_t* row_flags, int64_t* flag_list, float* flag_y,
                      int32_t flag_cap, int32_t* flag_count, float tau1, int blas_model, int32_t* counters,
                      int32_t* host_counts, const lshrs_sig_audit* audit, const lshrs_sig_opts* opts, void* stream) {
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || keys == nullptr || n < 0 || ldx < dim || flag_list == nullptr ||
      flag_count == nullptr || flag_cap <= 0 || !sig_shape_ok(num_bands, rows_per_band, dim))
    return LSHRS_E_BADARG;
  if (tie_list != nullptr && (tie_count == nullptr || tie_cap < 0)) return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  const int row_bytes = num_bands * g.bb;
  // (the second stage patches key bits with 32-bit atomics on the ALIGNED word around the byte: a word that straddles two
  //  rows, or the end of the buffer, shares its page with a byte that is ours, and the bits that are not ours go back as
  //  they came - key rows of any width, keys at any address)
  const bool narrow = sig_has_narrow_split(g);
  // short vectors of narrow hashers: the resident-image kernel (whole rows in registers: any dim % 4 == 0 with the replay)
  SigResident rs = sig_resident(num_bands, rows_per_band, dim);
  if (rs.on && dim % kKTile != 0 && blas_model == 0) rs.on = false;   // (only the replaying stage 2 masks a row's end)
  if (!sig_has_split(g) && !narrow && !rs.on) return LSHRS_E_TOOLARGE;
  const int64_t wgs = ((n + 127) / 128 + 7) / 8 * 8 * g.cb;          // (the most any launch below makes)
  if (n >= ((int64_t)1 << 42) || wgs > 0x7fffffffLL || g.cb > 65535) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Opts o = read_opts(opts);
  const float* base = static_cast<const float*>(workspace);
  // (a partial last k-tile - dim % 32 != 0 - only with the replay: its stage 2 is the one that reads the chunks past a row's end as zero)
  // With the replaying stage 2 rows of any length (from 9 elements) at any 4-byte address: both stage-1 kernels fetch the row's
  // last dim % 4 elements with its last four and shift them into place; stage 2 adds the library's scalar tail behind its tiles
  // (sig_fix8_kernel<., GENERAL>); the LDS-DMA takes 4-byte aligned addresses.  (Without the replay - blas_model 0 - as
  // before: 16-byte chunks of 16-byte aligned rows.)
  const bool aligned = ldx < (1 << 20) &&
                       (blas_model != 0 ? (dim % 4 == 0 || dim >= 9) && (dim >= 32 || rs.on)
                                        : dim % 32 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0);
  if (!aligned) {  // the split pass is built for 16-byte chunks of 16-byte aligned rows; anything else takes the f32 pass (same keys)
    if (blas_model != 0) return LSHRS_E_BADARG;   // (the f32 kernel reports ties, it does not resolve them)
    return lshrs_sig_hash_batch_f32(X, n, ldx, workspace, num_bands, rows_per_band, dim, keys, tie_list, tie_cap,
                                    tie_count, tau, row_flags, opts, stream);
  }
  // stage 1: bf16 x 3 projections -> keys + list of the projections inside the stage-1 window (+ their values)
  SigArgs a{};
  a.X = X;
  a.n = n;
  a.ldx = ldx;
  a.dim = dim;
  a.ktiles = g.ktiles;
  a.ncb = g.cb;
  a.image = base + sig_t16_offset_floats(g);
  a.norms = base + sig_image_floats(g);
  a.norm_max = a.norms + sig_norm_floats(g);
  if (narrow) {   // the zero-padded 256-column image and its norms
    a.ncb = 1;
    a.image = base + sig_narrow_offset_floats(g);
    a.norms = a.image + sig_narrow_image_floats(g);
    a.norm_max = a.norms + 256;
  }
  const SigCompact cp = sig_compact(g, num_bands, rows_per_band);
  const SigCompactWs cw = cp.on ? sig_compact_ws(const_cast<float*>(base), g, cp) : SigCompactWs{};
  if (cp.on) {    // fewer column blocks with the bands' columns side by side (sig_compact)
    a.ncb = cp.ncb;
    a.image = cw.image;
    a.norms = cw.norms;
    a.norm_max = cw.norm_max;
    a.compact = 1;
    a.padcol = cw.padcol;
    a.bytetab = cw.bytetab;
    a.bpb = cp.bpb;
    a.band_bytes = g.bb;
    a.num_bands = num_bands;
  }
  a.keys = keys;
  a.row_bytes = row_bytes;
  a.vec_store = (row_bytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(keys) % 16) == 0);
  a.row_base = 0;
  a.tie_list = flag_list;
  a.flag_y = flag_y;
  a.tie_cap = flag_cap;
  a.tie_count = flag_count;
  if (tau1 > 0.f) {                 // a window of tau1 units of ||x|| ||p||, the caller's responsibility
    a.tau = tau1;
    a.tau_b = 0.f;
    a.wa = a.wb = a.norms;
    a.wamax = a.wbmax = a.norm_max;
  } else {                          // LSHRS_WINDOW_PROVEN: ||x_hi|| wa + ||x_mid|| wb (lshrs_sig_set_window)
    const SigWindow w = sig_window(base, g);
    a.tau = a.tau_b = 1.0f;
    a.wa = w.wa;
    a.wb = w.wb;
    a.wamax = narrow ? w.wamax + kNarrowMaxSlot : w.wamax;
    a.wbmax = narrow ? w.wbmax + kNarrowMaxSlot : w.wbmax;
    if (cp.on) {
      a.wa = cw.wa;
      a.wb = cw.wb;
      a.wamax = cw.wamax;
      a.wbmax = cw.wbmax;
    }
  }
  a.row_flags = row_flags;
  a.clock_probe = o.clock_probe;
  // sig16_kernel's workgroup: 256 rows, or 128 (two per CU) - sig16_half_rows
  const bool half = !rs.on && sig16_half_rows(g.ktiles, (n + 255) / 256 * a.ncb);
  const int wg_rows = half ? 128 : 256, wg_waves = half ? 4 : 8;
  const int64_t row_tiles = (n + wg_rows - 1) / wg_rows;
  // the audit sample (lshrs_sig_audit): one unit in `div` - a wave of sig16_kernel, a 32-row tile of sig16r_kernel
  int audit_n = 0;
  if (audit != nullptr && audit->struct_bytes >= sizeof(lshrs_sig_audit) && audit->list != nullptr && audit->vals != nullptr &&
      audit->slots > 0 && audit->target > 0 && blas_model != 0) {
    const int res_rows = 16 * res_rt(rs.nct, rs.kt);
    const int64_t units = rs.on ? (n + res_rows - 1) / res_rows : (row_tiles + 7) / 8 * 8 * a.ncb * wg_waves;
    int64_t div = units / audit->target;
    if (div < 1) div = 1;
    if ((units + div - 1) / div > audit->slots) div = (units + audit->slots - 1) / audit->slots;
    if (div <= 0x7fffffffLL) {
      a.audit_list = audit->list;
      a.audit_vals = audit->vals;
      a.audit_div = (int)div;
      a.audit_phase = (int)(audit->seed % (uint32_t)div);
      a.audit_seed = audit->seed;
      audit_n = (int)((units - 1 - a.audit_phase) / div + 1);           // units u < `units` with u % div == phase: every slot is written
    }
  }
  // BUCKETS (lshrs_sig_sort mode 1): sig16_kernel appends flagged and sampled projections by key column (sig_replay.hip takes
  // them from there): long rows, at most 1024 padded key columns, the replaying stage 2, the caller's scratch given
  const int padcols_all = row_bytes * 8;
  const bool buckets = !rs.on && blas_model != 0 && rows_per_band != 1 && o.sort != nullptr && o.sort->mode == 1 && o.sort->thr != nullptr &&
                       padcols_all <= kSortMaxCols && g.ktiles > kFixSlabShort && o.sort->cap / padcols_all >= 64;
  if (buckets) {
    a.tie_list = o.sort->list;
    a.flag_y = o.sort->y;
    a.flag_thr = o.sort->thr;
    a.col_count = o.sort->hist + (size_t)(o.sort->parity & 1) * kSortMaxCols;
    a.col_cap = o.sort->cap / padcols_all;
  }
  if (rs.on) {
    const SigCompactWs rw = sig_resident_ws(const_cast<float*>(base), g, num_bands, rows_per_band, rs);
    a.ncb = 1;
    a.image = rw.image;
    a.norms = rw.norms;
    a.norm_max = rw.norm_max;
    a.compact = 1;
    a.padcol = rw.padcol;
    a.bytetab = rw.bytetab;
    a.bpb = num_bands;
    a.band_bytes = g.bb;
    a.num_bands = num_bands;
    a.vec_store = (row_bytes % 4 == 0) && ((reinterpret_cast<uintptr_t>(keys) % 4) == 0);
    if (tau1 > 0.f) {
      a.wa = a.wb = a.norms;
      a.wamax = a.wbmax = a.norm_max;
    } else {
      a.wa = rw.wa;
      a.wb = rw.wb;
      a.wamax = rw.wamax;
      a.wbmax = rw.wbmax;
    }
    const int res_rows = 16 * res_rt(rs.nct, rs.kt);
    const int64_t tiles = (n + res_rows - 1) / res_rows;
    const int rwaves = res_waves(rs.nct, rs.kt);
#ifndef LSHRS_RES_GRID
#define LSHRS_RES_GRID 256        // workgroups of a full launch: one per CU (A/B builds: more, shorter ones - the hardware hands them to the CUs that finish first)
#endif
    const unsigned grid = (unsigned)(tiles < (int64_t)LSHRS_RES_GRID * rwaves ? (tiles + rwaves - 1) / rwaves : LSHRS_RES_GRID);
    const int rc = lshrs_launch_sig16r(a, rs.nct, rs.kt, grid, 64u * (unsigned)rwaves, s, o.ev[0], o.ev[1]);
    if (rc != 0) return rc;
  } else {
    const int rc = lshrs_launch_sig16(a, (unsigned)((row_tiles + 7) / 8 * 8 * a.ncb), cp.on, dim % kKTile != 0, half, s, o.ev[0], o.ev[1]);
    if (rc != 0) return rc;
  }
  // stage 2: the flagged projections, one by one
  FixArgs f{};
  f.X = X;
  f.ldx = ldx;
  f.dim = dim;
  f.ktiles = g.ktiles;
  f.prow = base + sig_rowmajor_offset_floats(g);
  f.norms = (cp.on || rs.on) ? base + sig_image_floats(g) : a.norms;       // (stage 2 works on padded column ids throughout)
  f.keys = keys;
  f.row_bytes = row_bytes;
  f.padcols = row_bytes * 8;
  f.flag_list = flag_list;
  f.flag_count = flag_count;
  f.flag_cap = flag_cap;
  f.row_base = 0;
  f.tie_list = tie_list;
  f.tie_cap = tie_cap;
  f.tie_count = tie_count;
  f.tau = tau > 0.f ? tau : 1.0f;
  f.tie_coef = tau > 0.f ? f.norms : sig_window(base, g).wt;      // (proven tie window: coefficient per column, factor 1)
  f.blas_model = blas_model;
  f.rows_per_band = rows_per_band;
  f.band_cols = 8 * g.bb;
  if (buckets) {
    f.sorted_list = o.sort->list;
    f.sorted_y = o.sort->y;
    f.flag_thr = o.sort->thr;
    f.col_count = a.col_count;
    f.col_cap = a.col_cap;
  }
  if (blas_model != 0) {
    f.tail_model = blas_model;
    f.tie_list = nullptr;
    f.flag_y = flag_y;
    f.partials = counters + LSHRS_SIG_COUNTERS;
    f.count_ties = 1;
    f.audit_list = audit_n > 0 ? a.audit_list : nullptr;
    f.audit_vals = a.audit_vals;
    f.audit_n = audit_n;
  }
  const int rc2 = lshrs_replay_stage2(f, counters, host_counts, o, s);
  if (rc2 != 0) return rc2;
  return -(int)hipGetLastError();
}

int lshrs_sig_hash_batch_split_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                                   int32_t rows_per_band, int32_t dim, uint8_t* keys, int64_t* tie_list,
                                   int32_t tie_cap, int32_t* tie_count, float tau, uint8_t* row_flags,
                                   int64_t* flag_list, int32_t flag_cap, int32_t* flag_count, float tau1,
                                   const lshrs_sig_opts* opts, void* stream) {
  return split_pass(X, n, ldx, workspace, num_bands, rows_per_band, dim, keys, tie_list, tie_cap, tie_count, tau,
                    row_flags, flag_list, nullptr, flag_cap, flag_count, tau1, 0, nullptr, nullptr, nullptr, opts, stream);
}

int lshrs_sig_hash_batch_split_replay_f32(const float* X, int64_t n, int64_t ldx, const void* workspace,
                                          int32_t num_bands, int32_t rows_per_band, int32_t dim, uint8_t* keys,
                                          int32_t* counters, float tau, uint8_t* row_flags, int64_t* flag_list,
                                          float* flag_y, int32_t flag_cap, float tau1, int32_t blas_model,
                                          int32_t* host_counts, const lshrs_sig_audit* audit, const lshrs_sig_opts* opts,
                                          void* stream) {
  const bool resident = sig_resident(num_bands, rows_per_band, dim).on;
  // dim % 4 elements of scalar tail: modelled from 9 elements up, model 1 / 2 = how the build compiles them (sig_fixany_kernel);
  // whole groups of four: both builds sum alike, model 1
  const bool tail_ok = dim % 4 != 0 && dim >= 9 && rows_per_band >= 2 && (blas_model == 1 || blas_model == 2);
  // bands of ONE row: the host calls sdot - modelled for both builds at every length (stage 2: sig_fixany_kernel)
  const bool one_row_ok = rows_per_band == 1 && (blas_model == 1 || blas_model == 2);
  if (!tail_ok && !one_row_ok && (blas_model != 1 || dim % 4 != 0)) return LSHRS_E_BADARG;
  const int body = dim & ~3;
  if ((dim < 32 && !resident) || (body % 8 != 0 && body > 4096) || counters == nullptr) return LSHRS_E_BADARG;
  return split_pass(X, n, ldx, workspace, num_bands, rows_per_band, dim, keys, nullptr, 0, counters + 0, tau, row_flags,
                    flag_list, flag_y, flag_cap, counters + 1, tau1, blas_model, counters, host_counts, audit, opts, stream);
}

}  // extern "C"
