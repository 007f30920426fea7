// lshrs_common.h - what the translation units of liblshrs_hip.so share: vector types, the geometry of the hyperplane
// workspace (lshrs_sig_pack_projections builds it, every signature kernel reads it), the argument blocks of the kernels,
// the host BLAS's row kinds and reduction as device helpers, and the entry points one unit offers another.
// Everything here is inline / constexpr (no state, nothing exported); kernels live in the .hip files.
#ifndef LSHRS_COMMON_H
#define LSHRS_COMMON_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#include "lshrs_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
#define LSHRS_HIDDEN __attribute__((visibility("hidden")))

namespace lshrs {

// ------------------------------------------------------------------------------------------
// K1 geometry
// ------------------------------------------------------------------------------------------
constexpr int kKTile = 32;        // k per LDS tile; MFMA step s uses k = s (lanes 0-31) and 16+s (lanes 32-63)
constexpr int kRowsPerWave = 32;  // one 32-row MFMA tile per wave
constexpr int kSigWaves = 4;            // waves per workgroup of the f32 kernel: one per SIMD, two workgroups per CU
constexpr int kFragFloats = 64 * 4;  // one (column-tile, q) fragment block: 64 lanes x 4 floats = 1 KiB

struct SigGeom {
  int nt;        // 32-column tiles per workgroup (1, 2, 4 or 8)
  int cb;        // column blocks (grid.y)
  int ktiles;    // ceil(dim / 32)
  int bb;        // bytes per band
  int padcols;   // num_bands * bb * 8
  int tiles32;   // ceil(padcols / 32)
};

inline SigGeom sig_geom(int num_bands, int rows, int dim) {
  SigGeom g;
  g.bb = (rows + 7) / 8;
  g.padcols = num_bands * g.bb * 8;
  g.tiles32 = (g.padcols + 31) / 32;
  g.nt = g.tiles32 >= 8 ? 8 : (g.tiles32 > 2 ? 4 : (g.tiles32 > 1 ? 2 : 1));
  g.cb = (g.tiles32 + g.nt - 1) / g.nt;
  g.ktiles = (dim + kKTile - 1) / kKTile;
  return g;
}

inline int64_t sig_image_floats(const SigGeom& g) { return (int64_t)g.cb * g.ktiles * g.nt * 4 * kFragFloats; }
inline int64_t sig_norm_floats(const SigGeom& g) { return (int64_t)g.cb * g.nt * 32; }
inline int64_t sig_normmax_floats(const SigGeom& g) { return ((int64_t)g.cb + 3) & ~(int64_t)3; }

// "Fine" geometry: one 32-column tile per workgroup (NT = 1, one column block per tile).  Same arithmetic per
// projection; 8x more, 8x shorter workgroups than NT = 8.  Used where a launch cannot fill the chip with
// NT = 8 workgroups: small batches (a single query vector: 1/8 of the latency) and the partial last round of
// a large batch.  Its image follows the main one in the workspace; the per-column norms are shared.
inline SigGeom sig_fine_geom(const SigGeom& g) {
  SigGeom f = g;
  f.nt = 1;
  f.cb = g.tiles32;
  return f;
}
inline bool sig_has_fine(const SigGeom& g) { return g.nt > 1; }
inline int64_t sig_main_floats(const SigGeom& g) { return sig_image_floats(g) + sig_norm_floats(g) + sig_normmax_floats(g); }
inline int64_t sig_fine_floats(const SigGeom& g) {
  if (!sig_has_fine(g)) return 0;
  const SigGeom f = sig_fine_geom(g);
  return sig_image_floats(f) + sig_normmax_floats(f);
}
// Split-precision first pass: the fragment image with every hyperplane entry as two bf16 values (hi = bf16(p),
// mid = bf16(p - hi)) in 16x16x32 fragment order - same byte size as the f32 image; wide geometry (NT = 8) only.
inline bool sig_has_split(const SigGeom& g) { return g.nt == 8; }
inline int64_t sig_t16_offset_floats(const SigGeom& g) { return sig_main_floats(g) + sig_fine_floats(g); }
// "Narrow" hashers - 128 to 255 key columns, e.g. the reference's default num_perm = 128 - take the split pass too: on
// a 16x16x32 fragment image padded with zero hyperplanes to the 256 columns sig16_kernel<2,8> works on (a zero
// column gives y = +0: never flagged, bit 0, and its key bytes lie beyond row_bytes and are not stored).  Half the
// matrix work is wasted and it is still 1.5x the exact-f32 kernel.  Own image + 256 norms + their maximum.
// (128 .. 224 key columns: four to seven 32-column tiles - the f32 kernel's geometry is then one or two column blocks of
// NT = 4, the narrow image one block of 256 either way; its window maxima live in the last slot of the maxima arrays.)
inline bool sig_has_narrow_split(const SigGeom& g) { return g.nt < 8 && g.padcols >= 128; }
constexpr int kNarrowMaxSlot = 3;           // sig_normmax_floats(g) >= 4 and a narrow hasher has at most two column blocks
inline int64_t sig_narrow_offset_floats(const SigGeom& g) { return sig_main_floats(g) + sig_fine_floats(g); }
inline int64_t sig_narrow_image_floats(const SigGeom& g) { return (int64_t)g.ktiles * 8 * 4 * kFragFloats; }
// Stage 2 reads whole hyperplanes: a plain row-major copy P'[padded column][32 * ktiles] (zero rows / zero tail), so
// that a k-tile of a column is ONE 128-byte line (in the fragment image it is eight 16-byte pieces of eight lines).
inline int64_t sig_rowmajor_floats(const SigGeom& g) { return (int64_t)g.cb * g.nt * 32 * g.ktiles * kKTile; }
inline int64_t sig_rowmajor_offset_floats(const SigGeom& g) {
  return sig_main_floats(g) + sig_fine_floats(g) + (sig_has_split(g) ? sig_image_floats(g) : 0) +
         (sig_has_narrow_split(g) ? sig_narrow_image_floats(g) + 256 + 4 : 0);
}
// Window block (lshrs_sig_set_window): per padded column the coefficients of the PROVEN stage-1 window of the split pass
// (wa, wb: |y1 - y_host| <= ||x_hi|| wa + ||x_mid|| wb) and of the proven tie window of the f32 chain (wt), each zero-padded
// to at least 256 columns (the narrow split image), followed by their maxima per column block of the main geometry and,
// for wt, per 32-column tile of the fine geometry.
inline int64_t sig_window_offset_floats(const SigGeom& g) { return sig_rowmajor_offset_floats(g) + sig_rowmajor_floats(g); }
inline int64_t sig_window_cols(const SigGeom& g) { const int64_t c = (int64_t)g.cb * g.nt * 32; return c < 256 ? 256 : c; }
inline int64_t sig_window_floats(const SigGeom& g) {
  return 3 * sig_window_cols(g) + 3 * sig_normmax_floats(g) + sig_normmax_floats(sig_fine_geom(g));
}
struct SigWindow { const float *wa, *wb, *wt, *wamax, *wbmax, *wtmax, *wtmax_fine; };
inline SigWindow sig_window(const float* base, const SigGeom& g) {
  SigWindow w;
  const float* p = base + sig_window_offset_floats(g);
  const int64_t wc = sig_window_cols(g), cbp = sig_normmax_floats(g);
  w.wa = p; w.wb = p + wc; w.wt = p + 2 * wc;
  w.wamax = p + 3 * wc; w.wbmax = w.wamax + cbp; w.wtmax = w.wbmax + cbp; w.wtmax_fine = w.wtmax + cbp;
  return w;
}
// COMPACT column blocks of the split pass.  The padded layout gives every band 8 * ceil(rows / 8) columns - the layout of the
// keys - so a band of 10 rows wastes 6 of 16 columns and a band of 4 rows half of them, in matrix work as in fragments.
// Where that costs whole 256-column blocks (20 x 10: 320 padded columns = two blocks, 200 real ones = one) stage 1 runs
// on an image of its own: the key columns of whole bands side by side, `bpb` = 256 / rows bands per block (a band never
// straddles two blocks), the block's tail zero.  Only stage 1 knows: its list entries carry padded column ids (table
// `padcol`), its keys leave through a byte table (`bytetab`: source bit and mask of every key byte of the block), and it
// reads copies of the norms and window coefficients in its own order.  At the end of the workspace:
//   image_c [ncb * ktiles * 8192] | norms_c [ncb * 256] | norm_max_c [4..] | wa_c | wb_c [ncb * 256 each] | wamax_c | wbmax_c [4..]
//   | padcol int32 [ncb * 256] | bytetab int32 [ncb * 256 * 2]
struct SigCompact { bool on; int bpb; int ncb; };
inline SigCompact sig_compact(const SigGeom& g, int num_bands, int rows) {
  SigCompact c{false, 0, 0};
  if (g.nt != 8 || rows > 128) return c;
  c.bpb = 256 / rows;
  c.ncb = (num_bands + c.bpb - 1) / c.bpb;
  c.on = c.ncb < g.cb;
  return c;
}
inline int64_t sig_pad4(int64_t v) { return (v + 3) & ~(int64_t)3; }
struct SigCompactWs { float *image, *norms, *norm_max, *wa, *wb, *wamax, *wbmax; int *padcol, *bytetab; };
inline int64_t sig_compact_floats(const SigGeom& g, const SigCompact& c) {
  if (!c.on) return 0;
  return (int64_t)c.ncb * g.ktiles * 8192 + 3 * (int64_t)c.ncb * 256 + 3 * sig_pad4(c.ncb) + (int64_t)c.ncb * 256 * 3;
}
// RESIDENT image of sig16r_kernel (short vectors, at most 256 key columns): ONE compact column block - every band's rows
// side by side, `nct` 16-column tiles of it live - over `kt` (2, 4 or 8) k-tiles, zero beyond dim; same tables and copies as a
// compact block (SigCompactWs with ncb = 1, bpb = num_bands), behind the compact section.
struct SigResident { bool on; int nct; int kt; };
inline SigResident sig_resident(int num_bands, int rows, int dim) {
  SigResident r{false, 0, 0};
  const int64_t real = (int64_t)num_bands * rows;
  if (real > 256 || dim > 256 || dim < 8) return r;         // (any dim: the row's last dim % 4 elements are shifted into place)
  r.nct = (((int)real + 15) / 16 + 3) / 4 * 4;
  r.kt = dim <= 64 ? 2 : (dim <= 128 ? 4 : 8);
  r.on = r.nct * r.kt <= 64;                        // (the image - nct x kt x 2 KiB - and a wave's rows in flight must fit)
  return r;
}
// The compact column that sits at fragment position (column tile ct, row m of the tile) of the resident image.  sig16r_kernel
// computes P X^T: lane (r16, g) ends with columns m = 4 g + e of every column tile for ONE row and shifts their signs into a
// word value by value (ct ascending, e ascending: the first lands highest) - so that word IS 32 (a last group of four column
// tiles: 16) consecutive bits of the row's sign string: group k of eight column tiles = columns 128 k + bits g .. of the string.
__host__ __device__ inline int res_colmap(int nct, int ct, int m) {
  const int g = m >> 2, e = m & 3, k = ct >> 3;
  const int bits = 4 * (nct - 8 * k < 8 ? nct - 8 * k : 8);
  return 128 * k + bits * g + (bits - 1) - (4 * (ct & 7) + e);
}
inline int64_t sig_resident_floats(const SigResident& r) {
  return r.on ? (int64_t)r.kt * 8192 + 3 * 256 + 3 * 4 + 256 * 3 : 0;
}
inline int64_t sig_workspace_floats(const SigGeom& g, int num_bands, int rows, int dim) {
  return sig_window_offset_floats(g) + sig_window_floats(g) + sig_compact_floats(g, sig_compact(g, num_bands, rows)) +
         sig_resident_floats(sig_resident(num_bands, rows, dim));
}
inline SigCompactWs sig_compact_ws(float* base, const SigGeom& g, const SigCompact& c) {
  SigCompactWs w;
  float* p = base + sig_window_offset_floats(g) + sig_window_floats(g);
  w.image = p; p += (int64_t)c.ncb * g.ktiles * 8192;
  w.norms = p; p += (int64_t)c.ncb * 256;
  w.norm_max = p; p += sig_pad4(c.ncb);
  w.wa = p; p += (int64_t)c.ncb * 256;
  w.wb = p; p += (int64_t)c.ncb * 256;
  w.wamax = p; p += sig_pad4(c.ncb);
  w.wbmax = p; p += sig_pad4(c.ncb);
  w.padcol = reinterpret_cast<int*>(p); p += (int64_t)c.ncb * 256;
  w.bytetab = reinterpret_cast<int*>(p);
  return w;
}
inline SigCompactWs sig_resident_ws(float* base, const SigGeom& g, int num_bands, int rows, const SigResident& r) {
  SigCompactWs w;
  float* p = base + sig_window_offset_floats(g) + sig_window_floats(g) + sig_compact_floats(g, sig_compact(g, num_bands, rows));
  w.image = p; p += (int64_t)r.kt * 8192;
  w.norms = p; p += 256;
  w.norm_max = p; p += 4;
  w.wa = p; p += 256;
  w.wb = p; p += 256;
  w.wamax = p; p += 4;
  w.wbmax = p; p += 4;
  w.padcol = reinterpret_cast<int*>(p); p += 256;
  w.bytetab = reinterpret_cast<int*>(p);
  return w;
}
constexpr int64_t kRoundRows = 65536;       // rows one full round of workgroups covers: 256 CUs x 2 x 128 (or 1 x 256)

// Which geometry finishes a partial round (m < kRoundRows rows) sooner?  Cost model fitted to
// profiles/r01_fine_sweep.log (MI355X): NT-wide workgroups run in layers of one workgroup per CU, each layer
// taking about one tile time (~4.4 us per 32-deep k-tile + launch); the fine geometry is close to linear in the work.
inline bool sig_prefer_fine(const SigGeom& g, int64_t m) {
  const double layers = (double)(((m + 127) / 128 * g.cb + 255) / 256);
  const double t_main = layers * (4.4 * g.ktiles + 10.0);
  const double t_fine = 30.0 + 2.3e-5 * (double)m * g.tiles32 * g.ktiles;
  return t_fine < t_main;
}

// ------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------
struct SigArgs {
  const float* X;
  int64_t n;
  int64_t ldx;
  int dim;
  int ktiles;
  const float* image;
  const float* norms;
  const float* norm_max;  // per column block: max of norms
  // keys mode
  uint8_t* keys;
  int row_bytes;       // num_bands * bb
  int vec_store;       // 1: rows of keys may be written with aligned vector stores
  int64_t row_base;    // added to the row index reported in tie entries (launches over a row sub-range)
  int64_t* tie_list;
  int tie_cap;
  int* tie_count;
  float tau;
  uint8_t* row_flags;
  // stage 1 of the split pass (sig16_kernel): column blocks in the grid; optional stage-1 value per list entry
  int ncb;
  float* flag_y;
  // ... its window: |y1| <= tau ||x_hi|| wa[col] + tau_b ||x_mid|| wb[col] goes to stage 2.  Proven window: tau = tau_b = 1
  // and the coefficient arrays of lshrs_sig_set_window; a caller-chosen window of tau1 units: tau = tau1, tau_b = 0,
  // wa = the column norms.  (wamax / wbmax: per column block, for the wave-uniform screen.)
  const float* wa;
  const float* wb;
  const float* wamax;
  const float* wbmax;
  float tau_b;
  // ... compact column blocks (sig_compact): image, norms and coefficients above are in compact order; the list entries
  // and the keys go through these tables
  int compact;            // 0: the padded layout
  const int* padcol;      // [ncb * 256]
  const int* bytetab;     // [ncb * 256 * 2]
  int bpb;                // bands per block
  int band_bytes;
  int num_bands;
  // project mode
  float* Y;
  int64_t ldy;
  // diagnostics: when set, wave 0 of every workgroup stores {shader-clock ticks, 100 MHz ticks} of its main loop
  unsigned long long* clock_probe;
  // audit of what stage 1 does NOT flag (lshrs_sig_audit): one wave in `audit_div` (unit u = workgroup * 8 + wave for
  // sig16_kernel, the 32-row tile for sig16r_kernel; sampled when u % audit_div == audit_phase) leaves ONE of its
  // projections - chosen by a hash of (u, audit_seed) - in slot u / audit_div: the list entry, its stage-1 value and the
  // window it was compared with; -1 where the choice fell on a flagged projection, a padding column or a row past the end.
  // Every slot is written by every launch.
  int64_t* audit_list;
  float* audit_vals;
  int audit_div;
  int audit_phase;
  unsigned audit_seed;
  // BUCKETS (lshrs_sig_sort mode 1; sig16_kernel only): col_cap > 0 - flagged projections do not go to one list but to the
  // segment of their padded key column: entry `col_count[col]++` of tie_list / flag_y [col * col_cap ..) (tie_count still takes
  // the total), and the audit sample goes there too, marked by bit 62 of the entry, its window in flag_thr: stage 2 then finds
  // every group of eight entries under ONE hyperplane without a sort in between
  int* col_count;
  int col_cap;
  float* flag_thr;
};
constexpr int64_t kAuditBit = (int64_t)1 << 62;

__device__ __forceinline__ unsigned audit_hash(unsigned u, unsigned seed) {
  unsigned h = u * 0x9E3779B1u ^ (seed * 0x85EBCA6Bu + 0xC2B2AE35u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Ballot + deposit in one block.  The 64-lane compare result (VCC: low half = the 32 columns of row rho,
// high half = the same columns of row rho + 4) is written into the two lanes that own those output
// words with v_writelane_b32 (immediate lane select; this clang exposes no builtin for it).
// gfx940-family hazard: a VALU-written SGPR needs 2 wait states before the next VALU reads it, and hipcc
// pads nothing inside an asm statement — hence the s_nop 1.
__device__ __forceinline__ void deposit_positive(uint32_t& word, float y, int lane_lo, int lane_hi) {
  asm("v_cmp_lt_f32 vcc, 0, %1\n\t"
      "s_nop 1\n\t"
      "v_writelane_b32 %0, vcc_lo, %2\n\t"
      "v_writelane_b32 %0, vcc_hi, %3"
      : "+v"(word)
      : "v"(y), "n"(lane_lo), "n"(lane_hi)
      : "vcc");
}

__device__ __forceinline__ void deposit_abs_below(uint32_t& word, float y, float bound, int lane_lo, int lane_hi) {
  asm("v_cmp_lt_f32 vcc, |%1|, %2\n\t"
      "s_nop 1\n\t"
      "v_writelane_b32 %0, vcc_lo, %3\n\t"
      "v_writelane_b32 %0, vcc_hi, %4"
      : "+v"(word)
      : "v"(y), "v"(bound), "n"(lane_lo), "n"(lane_hi)
      : "vcc");
}



// ------------------------------------------------------------------------------------------
// Stage 2 of the split-precision pass, for every flagged (row, padded column).  Without the replay it re-evaluates
// the projection as the canonical f32 fmaf chain (the order of the f32 MFMA kernel and of oracle/chain_model.c:
// per 32-deep k-tile, step s multiplies k = 32t+s then k = 32t+16+s), corrects the key bit if stage 1 had it
// wrong, and reports the projection as a tie when |y| < tau * ||x|| * ||p||; with it, see sig_fix8_kernel.
// ------------------------------------------------------------------------------------------
struct FixArgs {
  const float* X;
  int64_t ldx;
  int dim;
  int ktiles;
  const float* prow;      // hyperplanes row-major: P'[padded column][ldp], ldp = 32 * ktiles
  const float* norms;
  const float* tie_coef;  // per padded column: the tie window is tau * ||x|| * tie_coef[col] (the norms, or the proven coefficients)
  uint8_t* keys;
  int row_bytes;
  int padcols;            // valid padded columns = row_bytes * 8
  const int64_t* flag_list;
  const int* flag_count;
  int flag_cap;
  int64_t row_base;
  int64_t* tie_list;
  int tie_cap;
  int* tie_count;
  float tau;
  int blas_model;         // sig_fix8_kernel<true>: which host-BLAS summation order the tie replay follows (1: see there)
  int rows_per_band;      // ... and what it needs to know which of the library's kernels computes a column: the band's rows
  int band_cols;          //     and its padded width (8 * band_bytes)
  const float* flag_y;    // optional: stage-1 value of every list entry (sig16_kernel stores it beside the entry)
  int count_ties;         // sig_fix8_kernel<true>: report the projections inside the tie window in partials[0] (else 0)
  int* partials;          // sig_fix8_kernel<true>: int32[kFixParts * gridDim.x], per workgroup: projections inside the tie
                          // window, flagged projections whose stage-1 sign differed from the host BLAS's, (float bits) the
                          // max over its flagged projections of |y1 - y_BLAS| in units of 2^-24 ||x|| ||p||, and the audit:
                          // projections audited, audited projections whose key bit is not the host's sign, (float bits)
                          // max over them of |y1 - y_BLAS| / the window stage 1 compared that projection with
  const int64_t* audit_list;   // the un-flagged projections stage 1 sampled (SigArgs::audit_list), audit_n slots, -1 = none:
  const float* audit_vals;     // replayed like the flagged ones behind them, nothing patched - only compared
  int audit_n;
  int tail_model;         // how the host compiles the dim % 4 elements behind the last group of four (1 / 2): sig_fixany_kernel, sig_fix8_kernel<., GENERAL>
  // the list BY KEY COLUMN (lshrs_sig_sort, SAMEP instantiations; BUCKETS, SigArgs::col_cap): every group of eight entries has
  // ONE column; sorted_list / sorted_y are the columns' segments (entries, their stage-1 values), col_count the entries stage 1
  // wanted per column (more than col_cap: the column overflowed - reported through *overflow, the pass is repeated with room),
  // flag_thr the window of the audit entries (bit 62 of an entry)
  const int64_t* sorted_list;
  const float* sorted_y;
  const int* col_count;
  int col_cap;
  const float* flag_thr;
  int* overflow;
};
constexpr int kFixParts = 6;

// The host BLAS's left-over rows (blas_row_kind != 0) multiply and add in TWO roundings: never contracted into an fma.
__device__ __forceinline__ float mul_then_add(float acc, float a, float b) {
#pragma clang fp contract(off)
  const float prod = a * b;
  return acc + prod;
}

// Which of OpenBLAS's sgemv_t kernels computes row j of a band of `rows` hyperplanes (lshrs_host.h, lshrs_tb_model_row_dot):
// 0 = the 8-lane fma kernel (rows in groups of four; also the zero rows a band is padded with), 1 = the 4x2 kernel (a
// pair of left-over rows: four unfused chains), 2 = the 4x1 kernel (a single left-over row, or the third: eight unfused chains).
__device__ __forceinline__ int blas_row_kind(int j, int rows) {
  const int r4 = rows & ~3;
  if (j < r4 || j >= rows) return 0;
  return ((rows & 3) == 1 || j - r4 == 2) ? 2 : 1;
}

// The library's reduction of a lane group's chains (lane = 8 sub + g; every lane takes part, sub = 0 holds the result):
// kinds 0 and 2: ((p0+p4) + (p1+p5)) + ((p2+p6) + (p3+p7)); kind 1 (chains in sub 0..3, mirrored in 4..7): (v0+v1) + (v2+v3).
__device__ __forceinline__ float blas_reduce(float pj, int kind, int lane) {
  const float o = __shfl(pj, (lane + 32) & 63);
  const float q = kind == 1 ? pj : pj + o;
  const float h = q + __shfl(q, (lane + 8) & 63);
  return h + __shfl(h, (lane + 16) & 63);
}

constexpr int kBlasBlockTiles = 128;     // the library consumes the vector in blocks of 4096 elements, each reduced on its own
static inline bool blas_general(int rows_per_band, int ktiles, int dim) {
  return (rows_per_band & 3) != 0 || ktiles > kBlasBlockTiles || dim % kKTile != 0;
}


// k-tile of x row g and of hyperplane column g go HBM/L2 -> LDS by LDS-DMA, landing as [k-tile][chunk][g] (one
// instruction = one k-tile of all eight projections = 1 KiB), and every lane then runs the canonical chain of ITS g
// from ds_read_b128s that hit eight distinct 16-byte slots (the eight lanes sharing a g read the same slot: broadcast).
// 1/8 of the waves, the same chain length per wave: one resident round of 1536 waves covers even a 524 288-row chunk's list.
constexpr int kFixG = 8;
#ifndef LSHRS_FIX_SLAB
#define LSHRS_FIX_SLAB 6
#endif
#ifndef LSHRS_FIX_GRID
#define LSHRS_FIX_GRID 1536
#endif
constexpr int kFixSlabG = LSHRS_FIX_SLAB;       // k-tiles per slab; two slabs are resident (one being read, one landing): 2 x 2 x 6 x 8 chunks x 8
                                   // projections x 16 B = 24 KiB of LDS per wave (a 768-deep row is four slabs)
constexpr int kFixGridG = LSHRS_FIX_GRID;    // 256 CUs x 6 resident single-wave workgroups
// SHORT vectors (at most four k-tiles: dim <= 128): one slab IS the whole row, so a slab of four (16 KiB per workgroup, ten
// workgroups per CU) and a grid of up to 2 048 let every group of a short list be in flight at once - the list of a 1 M x 128
// batch (15 k flagged + 4 k audited) takes one round trip instead of two and a half (17 -> ~9 us)
constexpr int kFixSlabShort = 4;
constexpr int kHalfMaxTiles = 11;   // sig16_kernel: vectors of up to this many k-tiles take 128-row workgroups, two per CU (sig_split.hip: sig16_half_rows)
constexpr int kHalfMaxGroups = 128; // ... and so do batches of up to this many 256-row workgroups (half of the CUs would have none)
constexpr int kFixGridShort = 2048;
static_assert(LSHRS_SIG_COUNTERS + kFixParts * kFixGridShort <= LSHRS_SIG_DEVICE_COUNTERS, "stage 2's per-workgroup slots must fit the counter block");
constexpr int kSortMaxCols = 1024;       // the column-sorted stage 2 (lshrs_sig_sort) takes hashers of up to this many padded key columns

// ---- the resident-image kernel's shape rules (sig16r.hip), also what the host sizes its launch by
#ifndef LSHRS_RES_RT2_MAX
#define LSHRS_RES_RT2_MAX 16       // (A/B builds: two row tiles per wave up to this many (column tile, k-tile) pairs)
#endif
// row tiles per wave: two where the accumulators (8 NCT RT registers) and the rows in flight (8 KT RT) leave room, else one
constexpr int res_rt(int nct, int kt) { return (nct * kt <= 16 || (nct * kt <= LSHRS_RES_RT2_MAX && kt <= 4)) ? 2 : 1; }
#ifndef LSHRS_RES_WAVES_32
#define LSHRS_RES_WAVES_32 12      // (A/B builds: waves per workgroup where 16 < NCT KT <= 32 - 16 = four per SIMD within 128 registers)
#endif
constexpr int res_waves(int nct, int kt) { return (res_rt(nct, kt) == 1 && nct * kt <= 32 && kt <= 4) ? LSHRS_RES_WAVES_32 : 8; }
constexpr int kResListCap = 64;                        // flagged projections a wave stages before it appends them
constexpr int res_wave_floats(int rt) { return 32 * rt + 144 * rt + 3 * kResListCap + 4; }   // windows a / b, sign words (9 per row), list (entry, y1), counter: a multiple of 16 B
template <int NCT, int KT>
constexpr int res_lds_floats() { return KT * NCT * 512 + 512 + 256 + 512 + res_waves(NCT, KT) * res_wave_floats(res_rt(NCT, KT)); }

struct Opts {            // the caller's lshrs_sig_opts, or all-null
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long* clock_probe = nullptr;
  const lshrs_sig_sort* sort = nullptr;
  int* done = nullptr;      // pinned host word (device-visible address) that receives `epoch` behind the counters
  int epoch = 0;
};
inline Opts read_opts(const lshrs_sig_opts* o) {
  Opts r;
  if (o != nullptr && o->struct_bytes >= offsetof(lshrs_sig_opts, sort)) {       // (a caller built against the struct without `sort`)
    r.ev[0] = static_cast<hipEvent_t>(o->ev_stage1_start);
    r.ev[1] = static_cast<hipEvent_t>(o->ev_stage1_stop);
    r.ev[2] = static_cast<hipEvent_t>(o->ev_stage2_start);
    r.ev[3] = static_cast<hipEvent_t>(o->ev_stage2_stop);
    r.clock_probe = static_cast<unsigned long long*>(o->clock_probe);
    if (o->struct_bytes >= offsetof(lshrs_sig_opts, done_host) && o->sort != nullptr && o->sort->struct_bytes >= sizeof(lshrs_sig_sort) &&
        o->sort->list != nullptr && o->sort->y != nullptr && o->sort->hist != nullptr)
      r.sort = o->sort;
    if (o->struct_bytes >= sizeof(lshrs_sig_opts) && o->done_host != nullptr) {
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, o->done_host, 0) == hipSuccess && dp != nullptr) {
        r.done = static_cast<int*>(dp);
        r.epoch = o->done_epoch;
      }
    }
  }
  return r;
}
inline bool sig_shape_ok(int32_t num_bands, int32_t rows, int32_t dim) {
  if (num_bands <= 0 || rows <= 0 || dim <= 0) return false;
  const int64_t padcols = (int64_t)num_bands * ((rows + 7) / 8) * 8;
  return padcols <= (1 << 21);
}

}  // namespace lshrs

// ---- entry points one translation unit offers another (hidden: not part of the C ABI)
LSHRS_HIDDEN int lshrs_launch_sig16(const lshrs::SigArgs& a, unsigned grid, bool compact, bool partial, bool half, hipStream_t s, hipEvent_t start,
                                    hipEvent_t stop);
LSHRS_HIDDEN int lshrs_launch_sig16r(const lshrs::SigArgs& a, int nct, int kt, unsigned grid, unsigned block, hipStream_t s,
                                     hipEvent_t start, hipEvent_t stop);
LSHRS_HIDDEN int lshrs_replay_stage2(const lshrs::FixArgs& f, int32_t* counters, int32_t* host_counts, const lshrs::Opts& o, hipStream_t s);
LSHRS_HIDDEN int lshrs_sort_u64_rows(uint64_t* items, int q, int64_t cpad, hipStream_t s);     // rerank.hip: K3's global network
LSHRS_HIDDEN uint32_t lshrs_flags_sig16(void);
LSHRS_HIDDEN uint32_t lshrs_flags_sig16r(void);
LSHRS_HIDDEN uint32_t lshrs_flags_replay(void);
LSHRS_HIDDEN uint32_t lshrs_flags_query(void);

#endif
