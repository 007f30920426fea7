// lshrs_hip.hip — gfx950 (MI355X / CDNA4) kernels + C ABI for the lshrs hot path.
//
//   K1  sig16_kernel      stage 1 of the split-precision signature pass (bf16 x 3 on v_mfma_f32_16x16x32_bf16, proven window)
//       sig16r_kernel     ... for short vectors: fragment image resident in LDS, rows straight into registers
//       sig_fix8_kernel   stage 2: the host BLAS's summation order replayed for every flagged (and audited) projection
//       sig_fixany_kernel ... with plain loads: any vector length, any 4-byte row address, one-row bands
//       sig_small_kernel  a query vector or a handful: every projection the replayed value, one round trip
//       sig_kernel        the exact-f32 pass (v_mfma_f32_32x32x2_f32, ballot bit-pack)
//   K2  cosine_kernel     gather + dot + norm cosine of candidates against a query
//   K3  topk_kernel       per-query descending order (LDS bitonic network)
//   + small helpers (hyperplane re-layout, row gather, key patch scatter)
//
// Written for wave64 / v_mfma_f32_32x32x2_f32 / 160 KiB LDS; there is no other target.
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.

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

namespace {

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
  if (real > 256 || dim > 256 || dim < 8 || dim % 4 != 0) return r;
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
// Hyperplane re-layout.  image[cb][kt][jt][q][lane][r] = P'[col = (cb*NT + jt)*32 + (lane&31)]
//                                                          [k   = kt*32 + 16*(lane>>5) + 4*q + r]
// where P' is P with every band padded to 8*B columns (zero rows) and k padded to 32 (zeros).
// One (jt, q) block is exactly what one ds_read_b128 per lane hands to four MFMA steps.
// ------------------------------------------------------------------------------------------
__global__ void pack_image_kernel(const float* __restrict__ P, int num_bands, int rows, int dim, int bb, int nt,
                                  int ktiles, int64_t chunks, f32x4* __restrict__ image) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= chunks) return;
  const int lane = (int)(c & 63);
  const int q = (int)((c >> 6) & 3);
  int64_t t = c >> 8;
  const int jt = (int)(t % nt);
  t /= nt;
  const int kt = (int)(t % ktiles);
  const int cb = (int)(t / ktiles);
  const int col = (cb * nt + jt) * 32 + (lane & 31);
  const int band = col / (bb * 8);
  const int bit = col % (bb * 8);
  const int k0 = kt * kKTile + 16 * (lane >> 5) + 4 * q;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (band < num_bands && bit < rows) {
    const float* src = P + ((int64_t)band * rows + bit) * dim;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (k0 + r < dim) v[r] = src[k0 + r];
  }
  image[c] = v;
}

__device__ __forceinline__ uint16_t bf16_rne_bits(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (finite inputs)
  return (uint16_t)(u >> 16);
}

__global__ void pack_rowmajor_kernel(const float* __restrict__ P, int num_bands, int rows, int dim, int bb, int cols,
                                     int ldp, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)cols * ldp) return;
  const int col = (int)(t / ldp), k = (int)(t % ldp);
  const int band = col / (bb * 8), bit = col % (bb * 8);
  out[t] = (band < num_bands && bit < rows && k < dim) ? P[((int64_t)band * rows + bit) * dim + k] : 0.f;
}

__global__ void pack_norm_kernel(const float* __restrict__ P, int num_bands, int rows, int dim, int bb, int cols,
                                 float* __restrict__ norms) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  const int band = col / (bb * 8);
  const int bit = col % (bb * 8);
  double s = 0.0;
  if (band < num_bands && bit < rows) {
    const float* src = P + ((int64_t)band * rows + bit) * dim;
    for (int k = 0; k < dim; ++k) s += (double)src[k] * (double)src[k];
  }
  norms[col] = (float)sqrt(s);
}

__global__ void pack_normmax_kernel(const float* __restrict__ norms, int cols_per_block, int cb, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= cb) return;
  float m = 0.f;
  for (int c = 0; c < cols_per_block; ++c) m = fmaxf(m, norms[b * cols_per_block + c]);
  out[b] = m;
}

__global__ void fill_kernel(float* __restrict__ dst, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

// coefficient arrays over the key columns (band-major, num_bands x rows) -> padded columns (zero elsewhere)
__global__ void window_scatter_kernel(const float* __restrict__ ca, const float* __restrict__ cb_, const float* __restrict__ ct,
                                      int num_bands, int rows, int bb, int wcols, float* __restrict__ wa,
                                      float* __restrict__ wb, float* __restrict__ wt) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= wcols) return;
  const int band = col / (bb * 8), bit = col % (bb * 8);
  const bool live = band < num_bands && bit < rows;
  const int j = band * rows + bit;
  wa[col] = live ? ca[j] : 0.f;
  wb[col] = live ? cb_[j] : 0.f;
  wt[col] = live ? ct[j] : 0.f;
}

// Tables of the compact column blocks (sig_compact): per compact column its padded column id (-1: the zero tail of a
// block), per key byte of a block the bit of the block's 256-bit sign string it starts at and the mask of its live bits.
// (res_nct > 0: the resident image's order - position i holds compact column res_colmap(res_nct, i / 16, i % 16).)
__global__ void compact_tables_kernel(int num_bands, int rows, int bb, int bpb, int ncb, int* __restrict__ padcol,
                                      int* __restrict__ bytetab, int res_nct = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncb * 256) return;
  const int blk = i >> 8;
  int cc = i & 255;
  {
    const int pc = res_nct > 0 ? ((cc >> 4) < res_nct ? res_colmap(res_nct, cc >> 4, cc & 15) : 256) : cc;
    const int band = blk * bpb + pc / rows, bit = pc % rows;
    padcol[i] = (pc < bpb * rows && band < num_bands) ? band * bb * 8 + bit : -1;
  }
  const int bl = cc / bb, q = cc % bb;                        // key byte cc of the block: byte q of its band bl
  const bool live = bl < bpb && blk * bpb + bl < num_bands && 8 * q < rows;
  const int nbits = live ? (rows - 8 * q < 8 ? rows - 8 * q : 8) : 0;
  bytetab[2 * i] = live ? bl * rows + 8 * q : 0;
  bytetab[2 * i + 1] = (1 << nbits) - 1;
}

// a per-padded-column array in the order of the compact blocks (0 in a block's tail)
__global__ void compact_gather_kernel(const float* __restrict__ src, const int* __restrict__ padcol, int n,
                                      float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = padcol[i] >= 0 ? src[padcol[i]] : 0.f;
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
};

__device__ __forceinline__ unsigned audit_hash(unsigned u, unsigned seed) {
  unsigned h = u * 0x9E3779B1u ^ (seed * 0x85EBCA6Bu + 0xC2B2AE35u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}


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

// K1f: the exact-f32 signature pass.  MODE 0: keys only, 1: keys + tie list, 2: raw projections (diagnostic).
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
  int tail_model;         // sig_fixany_kernel: how the host compiles the dim % 4 elements behind the last group of four (1 / 2)
  // the list sorted by padded column (lshrs_sig_sort, SAMEP instantiations): every group of eight entries has ONE column, runs
  // padded to whole groups with -1; sorted_y: the entries' stage-1 values in that order; sorted_count: entries incl. padding
  const int64_t* sorted_list;
  const float* sorted_y;
  const int* sorted_count;
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

__device__ __forceinline__ void fix_chain_tile(const f32x4 (&p4)[2][4], const f32x4 (&x4)[2][4], float& acc, float& ss) {
#pragma unroll
  for (int sstep = 0; sstep < 16; ++sstep) {
    const int q = sstep >> 2, r = sstep & 3;
    acc = __builtin_fmaf(x4[0][q][r], p4[0][q][r], acc);
    acc = __builtin_fmaf(x4[1][q][r], p4[1][q][r], acc);
    ss = __builtin_fmaf(x4[0][q][r], x4[0][q][r], ss);
    ss = __builtin_fmaf(x4[1][q][r], x4[1][q][r], ss);
  }
}

// EIGHT flagged projections per wave.  One wave per projection (the first version of this kernel) has all 64 lanes
// issue the same 2 x dim dependent fmas: at ~5 700 flagged projections per 262 144-row chunk that is 5-6 waves per
// SIMD x 6 k issue cycles, i.e. the kernel is bound by redundant VALU issue (measured 33 us per chunk).  Here lane
// (sub, g) = (lane >> 3, lane & 7) works for projection g of the wave's group: the eight 16-byte chunks (sub) of a
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
constexpr int kFixGridShort = 2048;
static_assert(LSHRS_SIG_COUNTERS + kFixParts * kFixGridShort <= LSHRS_SIG_DEVICE_COUNTERS, "stage 2's per-workgroup slots must fit the counter block");
static_assert(LSHRS_SIG_COUNTERS + kFixParts * kFixGridG <= LSHRS_SIG_DEVICE_COUNTERS, "stage 2's per-workgroup slots must fit the counter block");
//
// REPLAY: the tie-break on the device.  Every flagged projection gets the sign of the value the HOST BLAS computes for
// it - the reference's `projection @ vector` (lshrs/hash/lsh.py:200) - and only that value is computed, by replaying
// that library's summation order: blas_model 1 = eight interleaved single-rounded fma chains
// p_j = sum over k = j (mod 8) of a_k x_k, j = 0..7, reduced as ((p0+p4) + (p1+p5)) + ((p2+p6) + (p3+p7)) - the
// 8-lane AVX kernel + vextractf128 / vhaddps / vhaddps of OpenBLAS's sgemv_t (Haswell, Zen and SkylakeX builds;
// found by search, tools/blas_order/, and checked bit for bit against `P_band @ x` of the running process before a
// hasher uses it: lshrs_amd/_hostblas.py).  The eight lanes (sub) that serve one projection each own one p_j, four fmas
// per k-tile from the slab in LDS: 96 steps for a 768-deep row where the canonical chain walks 768.  No tie list, no host.
// That kernel takes a band's rows four at a time; the rows_per_band % 4 rows left over and vectors longer than the
// library's 4096-element block are the GENERAL variant's (blas_row_kind, mul_then_add, kBlasBlockTiles).
//
// The slabs are double-buffered across the whole list: while slab u is read, slab u + 1 - the next slab of the same
// eight projections or the first slab of the wave's next eight - is landing (2 x kFixSlabG LDS-DMAs per slab, always
// exactly that many, so the waits are counted: "all but the youngest 2 x kFixSlabG").
// GENERAL (REPLAY only): bands whose rows are not a multiple of four, vectors longer than one block of the library and
// vectors that are not whole 32-deep k-tiles - the lanes look up their column's kernel kind, kind-1 lanes walk both halves
// of every 8-element step, the partial sums are reduced and added up at every block boundary, chunks past the row's end
// are fetched from its start and read as zero, and a vector of 8 m + 4 elements gives its first four to the low lanes
// before the tiles begin AT the fifth (the library's order: lshrs_tb_model_row_dot).  The common shapes (16 x 16 x 768 ...)
// keep the plain loop.
// SAMEP (REPLAY only; round 5): the list comes SORTED BY COLUMN (fix_sort_* below), every group of eight entries shares one
// hyperplane - its row is fetched ONCE per slab (one LDS-DMA of SLAB x 128 bytes by 8 SLAB lanes) and read by all eight entries
// from the same LDS words, instead of eight times from L2: the x rows are then the only stream (72.6 against 98.7 us per 115 k
// entries at 768-d, profiles/r03_stage2_streams.log).  The audit sample is not sorted: it keeps the plain instantiation.
template <bool REPLAY, bool GENERAL = false, int SLAB = kFixSlabG, bool SAMEP = false>
__global__ __launch_bounds__(64) void sig_fix8_kernel(const FixArgs a) {
  static_assert(!SAMEP || (REPLAY && SLAB * 8 <= 64), "the shared hyperplane slab is one LDS-DMA of the wave");
  __shared__ __attribute__((aligned(16))) f32x4 xs[2][SLAB * 8 * kFixG];
  __shared__ __attribute__((aligned(16))) f32x4 ps[2][SAMEP ? 64 : SLAB * 8 * kFixG];
  const int lane = threadIdx.x, g = lane & (kFixG - 1), sub = lane >> 3;
  const int shh = sub >> 2, sq = sub & 3;           // this lane's chunk of every k-tile: k = 32 t + 16 shh + 4 sq + 0..3
  const int64_t* __restrict__ list = SAMEP ? a.sorted_list : a.flag_list;
  const float* __restrict__ ylist = SAMEP ? a.sorted_y : a.flag_y;
  const int cnt = SAMEP ? *a.sorted_count : min(*a.flag_count, a.flag_cap);
  const int fgroups = (cnt + kFixG - 1) / kFixG;
  const int groups = fgroups + (!SAMEP && REPLAY && a.audit_list != nullptr ? (a.audit_n + kFixG - 1) / kFixG : 0);   // audit groups behind the list's
  const size_t ldp = (size_t)a.ktiles * kKTile;
  const int head = GENERAL ? (a.dim & 4) : 0;                 // 8 m + 4 elements: the first four go ahead of the tiles
  const int body = GENERAL ? a.dim - head : a.ktiles * kKTile; // elements the tiles cover (from element `head` on)
  const int kt = GENERAL ? (body + kKTile - 1) / kKTile : a.ktiles;
  const int slabs = (kt + SLAB - 1) / SLAB;
  // statistics are kept per lane and leave the wave once, at the end (one atomic per flagged projection on a single
  // address serialises the whole kernel as soon as the list is long)
  int n_ties = 0, n_flips = 0, n_aud = 0, n_abad = 0;
  float max_dev = 0.f, max_ratio = 0.f;
  struct Item { int64_t row; int col; bool live; const float* xg; const float* pg; const float* xrow; int e; bool audit; };
  auto fetch = [&](int grp) {                       // list entry g of group grp (a short last group re-does its first entry, unused)
    Item it;
    it.audit = grp >= fgroups;                      // (uniform per wave: a group is the list's or the audit's)
    int64_t item;
    bool inlist;
    if (!it.audit) {
      it.e = grp * kFixG + g;
      inlist = it.e < cnt;
      item = list[inlist ? it.e : grp * kFixG];
      if (SAMEP && item < 0) {                       // padding behind a column's run: the group's first entry, fetched, never used
        inlist = false;
        item = list[grp * kFixG];
      }
    } else {
      it.e = (grp - fgroups) * kFixG + g;
      item = it.e < a.audit_n ? a.audit_list[it.e] : -1;
      inlist = item >= 0;
      if (!inlist) item = 0;                         // (an empty slot: row 0, column 0 - fetched, never used)
    }
    it.row = item >> 21;                            // relative to this launch's X / keys
    const int col_raw = (int)(item & ((1 << 21) - 1));
    it.live = inlist && col_raw < a.padcols;
    it.col = col_raw < a.padcols ? col_raw : 0;
    it.xrow = a.X + it.row * a.ldx;
    it.xg = it.xrow + head + 16 * shh + 4 * sq;
    it.pg = a.prow + (size_t)it.col * ldp + head + 16 * shh + 4 * sq;
#ifdef LSHRS_AB_FIX_SAME_P        // (A/B builds only: what a list sorted by column would make of the hyperplane stream - wrong keys by design)
    {
      const int c0 = (int)(a.flag_list[grp * kFixG < cnt ? grp * kFixG : 0] & ((1 << 21) - 1));
      it.pg = a.prow + (size_t)(c0 < a.padcols ? c0 : 0) * ldp + 16 * shh + 4 * sq;
    }
#endif
    return it;
  };
  auto issue = [&](const Item& it, int slab, int buf) {   // nothing lands in a VGPR; tiles past the row's end re-fetch its last
#pragma unroll
    for (int i = 0; i < SLAB; ++i) {
      const int t = slab * SLAB + i < kt ? slab * SLAB + i : kt - 1;
#ifndef LSHRS_AB_FIX_NO_X        // (A/B builds only: which of the two streams bounds stage 2 - wrong keys by design)
      const float* xsrc = it.xg + (size_t)t * kKTile;
      if (GENERAL && t * kKTile + 16 * shh + 4 * sq >= body) xsrc = it.xrow;      // past the row's end: never read, never used
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)xsrc, (LDS_AS void*)(xs[buf] + i * 64), 16, 0, 0);
#endif
#ifndef LSHRS_AB_FIX_NO_P
      if (!SAMEP)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(it.pg + (size_t)t * kKTile), (LDS_AS void*)(ps[buf] + i * 64),
                                         16, 0, 0);
#endif
    }
    if (SAMEP) {      // the group's ONE hyperplane: lane L brings chunk L & 7 of the slab's tile L >> 3 (lanes past the slab: its last)
      const int ti = (lane >> 3) < SLAB ? (lane >> 3) : SLAB - 1;
      const int t = slab * SLAB + ti < kt ? slab * SLAB + ti : kt - 1;
      const float* psrc = a.prow + (size_t)it.col * ldp + head + (size_t)t * kKTile + 4 * (lane & 7);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)psrc, (LDS_AS void*)ps[buf], 16, 0, 0);
    }
  };
  int grp = blockIdx.x;                             // uniform per wave
  if (grp < groups) {
  Item cur = fetch(grp);
  issue(cur, 0, 0);
  int buf = 0;
  for (;;) {
    const int nxt_grp = grp + (int)gridDim.x;
    const bool has_next = nxt_grp < groups;
    Item nxt = cur;
    if (has_next) nxt = fetch(nxt_grp);             // (older than every DMA issued below: it is here when they are)
    const int64_t row = cur.row;
    const int col = cur.col, e = cur.e;
    const bool live = cur.live;
    const int word = col >> 5, c = col & 31;
    float acc = 0.f, ss = 0.f, pj = 0.f, ytot = 0.f;
    bool blocks_done = false;
    const int kind = GENERAL ? blas_row_kind(col % a.band_cols, a.rows_per_band) : 0;
    if (GENERAL && head != 0) {
      // the library's 8-lane kernels take elements 0..3 with their low lanes (chains 0..3) before anything else; its 4-lane
      // kernel (kind 1: chains in sub 0..3, mirrored in 4..7) simply starts there.  fl(p x): the first link of either chain.
      const int hl = kind == 1 ? (sub & 3) : sub;
      if (hl < 4) {
        const float hx = cur.xrow[hl];
        pj = a.prow[(size_t)col * ldp + hl] * hx;
        if (sub < 4) ss = hx * hx;
      }
    }
    for (int sl = 0; sl < slabs; ++sl) {
      const int tiles = kt - sl * SLAB < SLAB ? kt - sl * SLAB : SLAB;
      bool more = true;
      if (sl + 1 < slabs) issue(cur, sl + 1, buf ^ 1);
      else if (has_next) issue(nxt, 0, buf ^ 1);
      else more = false;
#if defined(LSHRS_AB_FIX_NO_X) || defined(LSHRS_AB_FIX_NO_P)
      if (more) wait_vmcnt<SLAB>();
#else
      if (more) wait_vmcnt<SAMEP ? SLAB + 1 : 2 * SLAB>();        // this slab has landed, the next one is on its way
#endif
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (REPLAY) {
        // The library's value IS the reference's for every flagged projection, tie or not: the canonical chain (768
        // dependent fmas per lane) is not needed here, only p_sub: k = 32 t + 8 m + sub, m = 0..3 = chunk
        // 2 m + (sub >> 2), element sub & 3.  (ss: this lane's share of ||x||^2, for the tie statistics only.)
        const float* xf = reinterpret_cast<const float*>(xs[buf]);
        const float* pf = reinterpret_cast<const float*>(ps[buf]);
        if (!GENERAL) {
#pragma unroll 3
          for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int o = ((t * 8 + 2 * m + (sub >> 2)) * kFixG + g) * 4 + (sub & 3);
              const int op = SAMEP ? (t * 8 + 2 * m + (sub >> 2)) * 4 + (sub & 3) : o;     // (shared slab: no entry index)
              const float xv = xf[o];
              pj = __builtin_fmaf(pf[op], xv, pj);
              ss = __builtin_fmaf(xv, xv, ss);
            }
          }
        } else {
          for (int t = 0; t < tiles; ++t) {
            const int kb0 = (sl * SLAB + t) * kKTile;      // first element of this tile, counted from `head`
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int o = ((t * 8 + 2 * m + (sub >> 2)) * kFixG + g) * 4 + (sub & 3);
              const int op = SAMEP ? (t * 8 + 2 * m + (sub >> 2)) * 4 + (sub & 3) : o;
              // past the row's end BOTH factors read as zero: what the fetch brought there is the neighbouring hyperplane's
              // (or the window block's) and may be anything - 0 * Inf would poison a column whose own value is finite
              const bool in = kb0 + 8 * m + sub < body;
              const float xv = in ? xf[o] : 0.f, pv = in ? pf[op] : 0.f;
              ss = __builtin_fmaf(xv, xv, ss);
              if (kind == 1) {              // chain l = sub & 3 takes k = 8 m + l, then k = 8 m + 4 + l
                const int o0 = ((t * 8 + 2 * m) * kFixG + g) * 4 + (sub & 3), o1 = o0 + kFixG * 4;
                const int p0 = SAMEP ? (t * 8 + 2 * m) * 4 + (sub & 3) : o0, p1 = SAMEP ? p0 + 4 : o1;
                const int kl = kb0 + 8 * m + (sub & 3);
                pj = mul_then_add(pj, kl < body ? pf[p0] : 0.f, kl < body ? xf[o0] : 0.f);
                pj = mul_then_add(pj, kl + 4 < body ? pf[p1] : 0.f, kl + 4 < body ? xf[o1] : 0.f);
              } else if (kind == 2) {
                pj = mul_then_add(pj, pv, xv);
              } else {
                pj = __builtin_fmaf(pv, xv, pj);
              }
            }
            const int tile = sl * SLAB + t + 1;      // (uniform: every lane of the wave is at the same k-tile)
            if ((tile % kBlasBlockTiles) == 0 && tile < kt) {
              const float sblk = blas_reduce(pj, kind, lane);
              ytot = blocks_done ? ytot + sblk : sblk;
              blocks_done = true;
              pj = 0.f;
            }
          }
        }
      } else {
#pragma unroll 2
        for (int t = 0; t < tiles; ++t) {
          f32x4 p4[2][4], x4[2][4];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              x4[hh][q] = xs[buf][(t * 8 + hh * 4 + q) * kFixG + g];
              p4[hh][q] = ps[buf][(t * 8 + hh * 4 + q) * kFixG + g];
            }
          fix_chain_tile(p4, x4, acc, ss);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab has been read before the one after next lands on it
      buf ^= 1;
    }
    float yb = 0.f;
    if (REPLAY) {       // (every lane takes part in the shuffles; the result is used by the sub = 0 lanes)
      // sub 0..3: p_sub + p_(sub+4); sub 0: q0 + q1, sub 2: q2 + q3; sub 0: (q0 + q1) + (q2 + q3)
      yb = blas_reduce(pj, kind, lane);
      if (GENERAL && blocks_done) yb = ytot + yb;
      float s2 = ss + __shfl(ss, (lane + 32) & 63);
      s2 += __shfl(s2, (lane + 8) & 63);
      ss = s2 + __shfl(s2, (lane + 16) & 63);
    }
    if (REPLAY && cur.audit) {
      // a projection stage 1 decided on its own: its key bit must be the sign of the host's value, and its stage-1 value
      // must lie within the window it was compared with.  Nothing is patched: a disagreement is for the caller to raise.
      if (sub == 0 && live) {
        const uint8_t kbyte = a.keys[row * (int64_t)a.row_bytes + (col >> 3)];
        ++n_aud;
        if ((yb > 0.f) != (((kbyte >> (col & 7)) & 1) != 0)) ++n_abad;
        const float y1 = a.audit_vals[2 * e], thr = a.audit_vals[2 * e + 1];
        if (thr > 0.f) {
          const float ratio = __builtin_fabsf(y1 - yb) / thr;
          if (ratio < __builtin_inff()) max_ratio = __builtin_fmaxf(max_ratio, ratio);
        }
      }
    } else if (sub == 0 && live) {
    uint8_t* kb = a.keys + row * (int64_t)a.row_bytes + (col >> 3);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(kb);
    unsigned int* w32 = reinterpret_cast<unsigned int*>(addr & ~(uintptr_t)3);
    const unsigned int bitmask = 1u << (8 * (unsigned)(addr & 3) + (col & 7));
    bool want = acc > 0.f;
    const bool have = (*kb >> (col & 7)) & 1;
    if (REPLAY) {
      want = yb > 0.f;                             // (0, -0 and NaN give 0, as `projections > 0` does: lsh.py:204)
      const float scale = sqrtf(ss) * a.norms[col];                        // ||x|| ||p||
      if (__builtin_fabsf(yb) < a.tau * sqrtf(ss) * a.tie_coef[col]) ++n_ties;   // statistics: projections inside the tie window
      if (want != have) ++n_flips;
      if (ylist != nullptr && scale > 0.f) {
        // the live margin of stage 1: how far its value was from the host BLAS's, in the units its window is given in
        const float dev = __builtin_fabsf(ylist[e] - yb) / (scale * 0x1p-24f);
        if (dev < __builtin_inff()) max_dev = __builtin_fmaxf(max_dev, dev);   // (NaN - a row flagged wholesale - drops out)
      }
    }
    if (want != have) {
      if (want) atomicOr(w32, bitmask);
      else atomicAnd(w32, ~bitmask);
    }
    if (!REPLAY && a.tie_list != nullptr) {
      const float thr = a.tau * sqrtf(ss) * a.tie_coef[col];
      if (__builtin_fabsf(acc) < thr) {
        const int slot = atomicAdd(a.tie_count, 1);
        if (slot < a.tie_cap) {
          a.tie_list[2 * (int64_t)slot] = (row + a.row_base) * 65536 + word;
          a.tie_list[2 * (int64_t)slot + 1] = (int64_t)(1u << c);
        }
      }
    }
    }   // sub == 0 && live
    if (!has_next) break;
    cur = nxt;
    grp = nxt_grp;
  }
  }   // grp < groups
  if (REPLAY) {
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {        // the results sit in lanes 0..7 (sub = 0)
      n_ties += __shfl_xor(n_ties, off);
      n_flips += __shfl_xor(n_flips, off);
      max_dev = __builtin_fmaxf(max_dev, __shfl_xor(max_dev, off));
      n_aud += __shfl_xor(n_aud, off);
      n_abad += __shfl_xor(n_abad, off);
      max_ratio = __builtin_fmaxf(max_ratio, __shfl_xor(max_ratio, off));
    }
    // One plain store per workgroup into its own slot; the launch behind this kernel folds the slots into the counters.
    // (Atomics on the three counters - 1 536 waves on one address each - were 28 of this kernel's 50 us at 22 k entries.)
    if (lane == 0) {
      int* p = a.partials + kFixParts * blockIdx.x;
      p[0] = a.count_ties ? n_ties : 0;
      p[1] = n_flips;
      p[2] = __float_as_int(max_dev);
      p[3] = n_aud;
      p[4] = n_abad;
      p[5] = __float_as_int(max_ratio);
    }
  }
}

// ---- The stage-1 list sorted by padded column (round 5), for sig_fix8_kernel<.., SAMEP>: a counting sort in three launches.
// kSortWgs workgroups take one contiguous slice of the list each; (1) per-workgroup histogram of the columns in LDS, stored
// column-major; (2) one workgroup: every column's total rounded up to whole groups of eight, scanned over the columns, then
// over the workgroups inside a column - the slot where each workgroup's entries of each column start - and -1 into the
// padding behind every column's run; (3) the slices once more: every entry to its column's next slot (LDS cursors).
constexpr int kSortWgs = 256, kSortThreads = 256, kSortMaxCols = 1024;

__global__ __launch_bounds__(kSortThreads) void fix_sort_hist_kernel(const int64_t* __restrict__ list, const int* __restrict__ count,
                                                                     int cap, int padcols, int* __restrict__ wg_hist) {
  __shared__ int hist[kSortMaxCols];
  for (int c = threadIdx.x; c < padcols; c += kSortThreads) hist[c] = 0;
  __syncthreads();
  const int cnt = min(*count, cap);
  const int per = (cnt + kSortWgs - 1) / kSortWgs;
  const int lo = blockIdx.x * per, hi = min(cnt, lo + per);
  for (int e = lo + threadIdx.x; e < hi; e += kSortThreads) {
    const int col = (int)(list[e] & ((1 << 21) - 1));
    if (col < padcols) atomicAdd(&hist[col], 1);      // (a column out of range - an entry stage 2 skips - is dropped here)
  }
  __syncthreads();
  for (int c = threadIdx.x; c < padcols; c += kSortThreads) wg_hist[(size_t)c * kSortWgs + blockIdx.x] = hist[c];
}

__global__ __launch_bounds__(kSortMaxCols) void fix_sort_scan_kernel(int* __restrict__ wg_hist, int padcols, int64_t* __restrict__ sorted,
                                                                      int* __restrict__ sorted_count) {
  __shared__ int tot[kSortMaxCols];
  const int c = threadIdx.x;
  int mine = 0;
  if (c < padcols)
    for (int w = 0; w < kSortWgs; ++w) mine += wg_hist[(size_t)c * kSortWgs + w];
  const int padded = (mine + kFixG - 1) / kFixG * kFixG;
  tot[c] = c < padcols ? padded : 0;
  __syncthreads();
  for (int off = 1; off < kSortMaxCols; off <<= 1) {      // inclusive scan over the columns
    const int v = c >= off ? tot[c - off] : 0;
    __syncthreads();
    tot[c] += v;
    __syncthreads();
  }
  if (c < padcols) {
    int at = tot[c] - padded;                             // where this column's run starts
    for (int w = 0; w < kSortWgs; ++w) {
      const int n = wg_hist[(size_t)c * kSortWgs + w];
      wg_hist[(size_t)c * kSortWgs + w] = at;
      at += n;
    }
    for (int e = at; e < tot[c]; ++e) sorted[e] = -1;     // at most seven
  }
  if (c == kSortMaxCols - 1) *sorted_count = tot[c];
}

__global__ __launch_bounds__(kSortThreads) void fix_sort_scatter_kernel(const int64_t* __restrict__ list, const float* __restrict__ y,
                                                                        const int* __restrict__ count, int cap, int padcols,
                                                                        const int* __restrict__ wg_hist, int64_t* __restrict__ sorted,
                                                                        float* __restrict__ sorted_y) {
  __shared__ int cursor[kSortMaxCols];
  for (int c = threadIdx.x; c < padcols; c += kSortThreads) cursor[c] = wg_hist[(size_t)c * kSortWgs + blockIdx.x];
  __syncthreads();
  const int cnt = min(*count, cap);
  const int per = (cnt + kSortWgs - 1) / kSortWgs;
  const int lo = blockIdx.x * per, hi = min(cnt, lo + per);
  for (int e = lo + threadIdx.x; e < hi; e += kSortThreads) {
    const int64_t item = list[e];
    const int col = (int)(item & ((1 << 21) - 1));
    if (col >= padcols) continue;
    const int at = atomicAdd(&cursor[col], 1);
    sorted[at] = item;
    if (y != nullptr) sorted_y[at] = y[e];
  }
}

// Tie entries of the f32 kernel, (row * 65536 + word, mask of up to 32 columns), unpacked into the stage-2 list format
// (row << 21 | padded column), one item per flagged column: what sig_fix8_kernel<true> takes.
__global__ void expand_ties_kernel(const int64_t* __restrict__ tie_list, const int* __restrict__ tie_count, int tie_cap,
                                   int padcols, int64_t* __restrict__ flag_list, int flag_cap, int* flag_count) {
  const int cnt = min(*tie_count, tie_cap);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += gridDim.x * blockDim.x) {
    const int64_t row = tie_list[2 * (int64_t)e] >> 16;
    const int word = (int)(tie_list[2 * (int64_t)e] & 0xFFFF);
    unsigned mask = (unsigned)tie_list[2 * (int64_t)e + 1];
    const int m = __popc(mask);
    if (m == 0) continue;
    int slot = atomicAdd(flag_count, m);
    while (mask != 0u) {
      const int c = __ffs(mask) - 1;
      mask &= mask - 1u;
      const int col = 32 * word + c;
      if (slot < flag_cap && col < padcols) flag_list[slot] = (row << 21) | (int64_t)col;
      else if (slot < flag_cap) flag_list[slot] = (row << 21) | (int64_t)((1 << 21) - 1);   // (skipped by stage 2: column out of range)
      ++slot;
    }
  }
}

// The tie replay for inputs the LDS-DMA form above does not take: vectors whose length is not a multiple of four (the
// library's scalar tail, lshrs_tb_model_row_dot: blas_model 1 as its SkylakeX build contracts it, 2 as its Haswell / Zen
// build leaves it), rows that are only 4-byte aligned.  Same lane roles - lane (sub, g) owns chain `sub` of list entry g of
// the wave's eight - with plain 4-byte loads: the eight lanes of an entry read 32 consecutive bytes of its row and of its
// hyperplane per step.  Only behind the f32 kernel (lshrs_sig_resolve_ties_replay_f32): a tie list is short, and the rate
// of this kernel (a few ns per entry) does not matter next to the pass in front of it.
__global__ __launch_bounds__(64) void sig_fixany_kernel(const FixArgs a) {
  const int lane = threadIdx.x, g = lane & (kFixG - 1), sub = lane >> 3;
  const int cnt = min(*a.flag_count, a.flag_cap);
  const int groups = (cnt + kFixG - 1) / kFixG;
  const size_t ldp = (size_t)a.ktiles * kKTile;
  const int body = a.dim & ~3, m3 = a.dim & 3;
  int n_ties = 0, n_flips = 0;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {     // (uniform per wave)
    const int e = grp * kFixG + g;
    const int64_t item = a.flag_list[e < cnt ? e : grp * kFixG];
    const int64_t row = item >> 21;
    const int col_raw = (int)(item & ((1 << 21) - 1));
    const bool live = e < cnt && col_raw < a.padcols;
    const int col = col_raw < a.padcols ? col_raw : 0;
    const float* __restrict__ xr = a.X + row * a.ldx;
    const float* __restrict__ pr = a.prow + (size_t)col * ldp;
    const int kind = blas_row_kind(col % a.band_cols, a.rows_per_band);
    float y = 0.f, ss = 0.f;
    if (a.rows_per_band == 1) {
      // A band of ONE row: NumPy calls sdot (lshrs_host.h, tb_model_sdot): the first n1 = dim & -32 elements through the
      // build's SIMD kernel (f32 result), the f32 products of the elements behind them summed one by one in a double, the
      // kernel's result added to that double, one rounding to f32.  Lane `sub` owns the chains c = sub + 8 j.
      const int n1 = a.dim & ~31;
      float kernel = 0.f;
      if (a.tail_model == 1) {          // SkylakeX build: 64-element steps on 64 chains, folded in halves; a last 32-element
        const int n64 = n1 & ~63;       // step onto the folded accumulators; those added in turn
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < n64; k0 += 64)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xv = xr[k0 + 8 * j + sub];
            acc[j] = __builtin_fmaf(pr[k0 + 8 * j + sub], xv, acc[j]);
            ss = __builtin_fmaf(xv, xv, ss);
          }
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = n64 > 0 ? acc[2 * u] + acc[2 * u + 1] : 0.f;
        if (n64 < n1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float xv = xr[n64 + 8 * u + sub];
            t[u] = __builtin_fmaf(pr[n64 + 8 * u + sub], xv, t[u]);
            ss = __builtin_fmaf(xv, xv, ss);
          }
        }
        const float v = ((t[0] + t[1]) + t[2]) + t[3];
        kernel = blas_reduce(v, 0, lane);    // lanes i + (i + 4), then (w0 + w1) + (w2 + w3)
      } else {                          // Haswell / Zen build: 32 chains, accumulators pairwise, lanes pairwise
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < n1; k0 += 32)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xv = xr[k0 + 8 * j + sub];
            acc[j] = __builtin_fmaf(pr[k0 + 8 * j + sub], xv, acc[j]);
            ss = __builtin_fmaf(xv, xv, ss);
          }
        float sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sj[j] = acc[j] + __shfl(acc[j], (lane + 32) & 63);   // a(2j) + a(2j+1): lanes l and l + 4
        const float v = (sj[0] + sj[1]) + (sj[2] + sj[3]);
        const float h = v + __shfl(v, (lane + 8) & 63);
        kernel = h + __shfl(h, (lane + 16) & 63);
      }
      if (n1 == 0) kernel = 0.f;
      double tail = 0.0;                // (every lane of the entry computes the same tail: at most 31 elements)
      for (int k = n1; k < a.dim; ++k) {
        const float xv = xr[k];
        tail += (double)mul_then_add(0.f, pr[k], xv);       // the f32 product, rounded on its own
        if (sub == 0) ss = __builtin_fmaf(xv, xv, ss);
      }
      y = (float)(tail + (double)kernel);
    } else
    for (int k0 = 0; k0 < body; k0 += 4096) {                      // the library's blocks (uniform trip count)
      const int kn = body - k0 < 4096 ? body - k0 : 4096;
      float pj = 0.f;
      if (kind == 1) {                                             // four unfused chains over k mod 4, mirrored in sub 4..7
        for (int k = k0 + (sub & 3); k < k0 + kn; k += 4) {
          const float xv = xr[k];
          pj = mul_then_add(pj, pr[k], xv);
          if (sub < 4) ss = __builtin_fmaf(xv, xv, ss);
        }
      } else {
        const int head = kn & 4;                                   // a block of 8 m + 4 elements: its first four go first
        if (head != 0 && sub < 4) {
          const float xv = xr[k0 + sub];
          pj = pr[k0 + sub] * xv;
          ss = __builtin_fmaf(xv, xv, ss);
        }
        for (int k = k0 + head + sub; k < k0 + kn; k += 8) {
          const float xv = xr[k];
          pj = kind == 0 ? __builtin_fmaf(pr[k], xv, pj) : mul_then_add(pj, pr[k], xv);
          ss = __builtin_fmaf(xv, xv, ss);
        }
      }
      const float sblk = blas_reduce(pj, kind, lane);              // (every lane takes part in the shuffles)
      y = k0 == 0 ? sblk : y + sblk;
    }
    if (m3 != 0 && a.rows_per_band != 1) {                         // the scalar tail (lshrs_tb_model_row_dot)
      const float a0 = pr[body], x0 = xr[body];
      const float a1 = m3 > 1 ? pr[body + 1] : 0.f, x1 = m3 > 1 ? xr[body + 1] : 0.f;
      const float a2 = m3 > 2 ? pr[body + 2] : 0.f, x2 = m3 > 2 ? xr[body + 2] : 0.f;
      if (sub == 0) ss = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, __builtin_fmaf(x2, x2, ss)));
      if (a.tail_model == 2) {                                     // nothing contracted
        float t = mul_then_add(0.f, a0, x0);
        if (m3 > 1) t = mul_then_add(t, a1, x1);
        if (m3 > 2) t = mul_then_add(t, a2, x2);
        y = mul_then_add(y, t, 1.0f);
      } else if (m3 == 1) {
        y = __builtin_fmaf(a0, x0, y);
      } else {
        float t = __builtin_fmaf(a0, x0, mul_then_add(0.f, a1, x1));
        if (m3 > 2) t = __builtin_fmaf(a2, x2, t);
        y = mul_then_add(y, t, 1.0f);
      }
    }
    float s2 = ss + __shfl(ss, (lane + 32) & 63);
    s2 += __shfl(s2, (lane + 8) & 63);
    s2 += __shfl(s2, (lane + 16) & 63);
    if (sub == 0 && live) {
      uint8_t* kb = a.keys + row * (int64_t)a.row_bytes + (col >> 3);
      const uintptr_t addr = reinterpret_cast<uintptr_t>(kb);
      unsigned int* w32 = reinterpret_cast<unsigned int*>(addr & ~(uintptr_t)3);
      const unsigned int bitmask = 1u << (8 * (unsigned)(addr & 3) + (col & 7));
      const bool want = y > 0.f;                                   // (0, -0 and NaN give 0: lsh.py:204)
      const bool have = (*kb >> (col & 7)) & 1;
      if (__builtin_fabsf(y) < a.tau * sqrtf(s2) * a.tie_coef[col]) ++n_ties;
      if (want != have) {
        ++n_flips;
        if (want) atomicOr(w32, bitmask);
        else atomicAnd(w32, ~bitmask);
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    n_ties += __shfl_xor(n_ties, off);
    n_flips += __shfl_xor(n_flips, off);
  }
  if (lane == 0) {
    int* p = a.partials + kFixParts * blockIdx.x;
    p[0] = a.count_ties ? n_ties : 0;
    p[1] = n_flips;
    p[2] = p[3] = p[4] = p[5] = 0;
  }
}

// Behind stage 2 of a replay pass: folds the per-workgroup statistics (nparts slots of 3 ints behind the
// LSHRS_SIG_COUNTERS counters: ties, sign flips, max deviation) into the counters, hands the counters to the host (pinned
// memory) and leaves the whole block zeroed for the next call: one small launch instead of a copy and a fill.
// Without host_counts the folded counters stay in the device block (the caller copies it).
constexpr int kExportThreads = 1024;      // one part or two per thread: the fold is one memory round trip deep, not nparts / 64
__global__ __launch_bounds__(kExportThreads) void export_counts_kernel(int* counters, int* host_counts, int nparts) {
  __shared__ int fold[kExportThreads / 64][kFixParts];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* parts = counters + LSHRS_SIG_COUNTERS;
  int ties = 0, flips = 0, dev = 0, aud = 0, abad = 0, ratio = 0;
  for (int i = tid; i < nparts; i += kExportThreads) {
    int* q = parts + kFixParts * i;
    ties += q[0];
    flips += q[1];
    dev = max(dev, q[2]);                            // (non-negative floats order like their bits)
    aud += q[3];
    abad += q[4];
    ratio = max(ratio, q[5]);
    q[0] = q[1] = q[2] = q[3] = q[4] = q[5] = 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    ties += __shfl_xor(ties, off);
    flips += __shfl_xor(flips, off);
    dev = max(dev, __shfl_xor(dev, off));
    aud += __shfl_xor(aud, off);
    abad += __shfl_xor(abad, off);
    ratio = max(ratio, __shfl_xor(ratio, off));
  }
  if (lane == 0) {
    fold[wave][0] = ties; fold[wave][1] = flips; fold[wave][2] = dev;
    fold[wave][3] = aud; fold[wave][4] = abad; fold[wave][5] = ratio;
  }
  __syncthreads();
  if (tid < LSHRS_SIG_COUNTERS) {
    ties = flips = dev = aud = abad = ratio = 0;
    for (int w = 0; w < kExportThreads / 64; ++w) {
      ties += fold[w][0];
      flips += fold[w][1];
      dev = max(dev, fold[w][2]);
      aud += fold[w][3];
      abad += fold[w][4];
      ratio = max(ratio, fold[w][5]);
    }
    int v = counters[tid];
    if (tid == 0) v += ties;
    if (tid == 2) v = max(v, dev);
    if (tid == 3) v += flips;
    if (tid == 4) v += aud;
    if (tid == 5) v += abad;
    if (tid == 6) v = max(v, ratio);
    if (host_counts != nullptr) {
      host_counts[tid] = v;
      counters[tid] = 0;
    } else {
      counters[tid] = v;
    }
  }
}

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
// ------------------------------------------------------------------------------------------
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
// grid-stride, the last nbytes % 16 bytes one by one.
__global__ __launch_bounds__(256) void copy_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                            const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, int tail) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
  if (blockIdx.x == 0 && tail > 256)
    for (int i = 256 + threadIdx.x; i < tail; i += 256) dst_tail[i] = src_tail[i];
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

// ------------------------------------------------------------------------------------------
// K1 on v_mfma_f32_16x16x32_bf16 ("T16").  Same algorithm, same bits, same LDS-DMA staging as the
// PIPE = 4 path of sig_kernel (x in full 128-byte lines, fragment ring of three 16 KiB stages), but on
// the 16x16x32 MFMA: at equal cycles per FLOP that shape draws less power, and on random operands the
// bf16 matrix pipes are power-capped long before they are issue-bound (tools/mfma_power_bench.hip:
// 32x32x16 runs at 1.5-1.6 GHz, 16x16x32 at 1.8-2.0 GHz = 1.2x the FLOP/s).
//   wave  = 64 rows (four 16-row tiles) x 256 columns (sixteen 16-column tiles) = 64 accumulator tiles of
//           4 registers; workgroup = 4 waves, one per SIMD, 256 rows; grid = ceil(n / 256) x column blocks.
//   stage = (k-tile t of 32, column half ch): 16 KiB of fragments = 8 column tiles x {hi, mid}, 96 MFMAs per
//           wave (16 cycles each) in two quarters of 48; x is read back and split once per k-tile.
//   image = image16[cb][t][ct 0..15][part][lane]: 16 bytes = the 8 bf16 of P'[col = 256 cb + 16 ct + (lane & 15)]
//           [k = 32 t + 8 (lane >> 4) + 0..7]: a stage is 16 KiB contiguous, staged by a linear LDS-DMA copy.
//   accumulator tile: column = lane & 15, row = 4 (lane >> 4) + register.
// ------------------------------------------------------------------------------------------
__global__ void pack_image_bf16_t16_kernel(const float* __restrict__ P, int num_bands, int rows, int dim, int bb,
                                           int ktiles, int64_t chunks, u16x8* __restrict__ image, int bpb = 0, int res_nct = 0) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= chunks) return;
  const int lane = (int)(c & 63);
  const int part = (int)((c >> 6) & 1);
  const int ct = (int)((c >> 7) & 15);
  const int64_t t = c >> 11;
  const int kt = (int)(t % ktiles);
  const int cb = (int)(t / ktiles);
  const int col = cb * 256 + ct * 16 + (lane & 15);
  int band = col / (bb * 8);
  int bit = col % (bb * 8);
  if (bpb > 0) {                                     // compact column blocks: bpb whole bands per block, no padding inside
    int cc = col & 255;
    if (res_nct > 0) cc = ct < res_nct ? res_colmap(res_nct, ct, lane & 15) : 256;     // the resident image's column order
    band = cc < bpb * rows ? cb * bpb + cc / rows : num_bands;
    bit = cc % rows;
  }
  const int k0 = kt * kKTile + 8 * (lane >> 4);
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (band < num_bands && bit < rows) {
    const float* src = P + ((int64_t)band * rows + bit) * dim;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (k0 + j < dim) {
        const float x = src[k0 + j];
        const uint16_t hi = bf16_rne_bits(x);
        const float hif = __uint_as_float((uint32_t)hi << 16);
        v[j] = part == 0 ? hi : bf16_rne_bits(x - hif);
      }
    }
  }
  image[c] = v;
}

// Stage 1 of the split-precision pass.  RT = 2 sixteen-row tiles per wave, W = 8 waves per workgroup (256 rows):
// 32 rows per wave, 128 accumulator AGPRs, at most 256 registers -> TWO waves per SIMD that share one fragment stage:
// each can issue MFMAs while the other sits in a DMA issue, a barrier or its VALU slices.  (Measured and dropped in
// round 1: 64 rows per wave / one wave per SIMD, a persistent variant, the 32x32x16 MFMA shape - DESIGN.md §5.)
// Grid: one dimension, blockIdx.x -> (row tile, column block) with the column blocks of one row tile eight workgroup
// ids apart: workgroups are dealt round-robin over the 8 XCDs, so the `cb` passes over the same 256 rows run on the
// SAME XCD at about the same time and the second one reads x from that XCD's L2 instead of HBM (config 5: 512 key
// columns = two column blocks).
#ifndef LSHRS_X_AUX
#define LSHRS_X_AUX 0              // cache policy of stage 1's x loads (A/B builds: 2 = nt)
#endif
constexpr int kS1ListCap = 8192;   // flagged projections a workgroup stages in LDS before its ONE global append
// COMPACT: the column blocks hold the bands' key columns side by side (sig_compact) - list entries and keys leave through
// the tables; a template parameter so that the padded layout's kernel is instruction for instruction what it was.
// PARTIAL: vectors that are not whole 32-element k-tiles (300-d, 100-d; dim % 4 == 0) - in the last k-tile the 16-byte
// chunks past a row's end are fetched from the tile's first chunk instead (never past the end of X) and read as zero.
template <bool COMPACT, bool PARTIAL = false>
__global__ __launch_bounds__(512, 1) void sig16_kernel(const SigArgs args) {
  constexpr int RT = 2, W = 8;
  constexpr int kWaveRows = 16 * RT;
  constexpr int kPP = 16 / W;                     // fragment pieces a wave stages per stage
  constexpr int kXPS = RT;                        // x pieces a wave stages per stage (2 RT per k-tile)
  constexpr int kE = 6 * RT;                      // MFMAs per eighth: 2 column tiles x 3 terms x RT row tiles
  constexpr int kSlices = 12 * RT;                // split slices per k-tile: 4 RT pairs x 3 steps
  constexpr int kPHalf = 16 * kFragFloats;        // floats of one fragment stage (16 blocks of 1 KiB)
  constexpr int kXTile = 256 * kKTile;            // floats of one x tile of the workgroup
  constexpr int kXWave = kWaveRows * kKTile;
  constexpr int kRingFloats = 3 * kPHalf + 3 * kXTile;
  static_assert(3 * kS1ListCap <= kRingFloats, "the epilogue's list stage reuses the ring");
  __shared__ __attribute__((aligned(16))) float lds[kRingFloats + 512 + 4];     // + two windows per row + list counters
  struct Bf16Pairs { bf16x2 p[4]; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  // blockIdx.x = ((group * ncb + cb) * 8 + xcd slot): row tile = group * 8 + slot
  const int ncb = args.ncb;
  const int bid = blockIdx.x;
  const int cb = (bid >> 3) % ncb;
  const int row_tile = ((bid >> 3) / ncb) * 8 + (bid & 7);
  // the audit sample of this wave (SigArgs::audit_list): slot au_slot (-1: not sampled), accumulator word au_rw = 8 rt + w,
  // lane au_lane, slot au_q of that word's eight values
  int au_slot = -1, au_rw = -1, au_lane = 0, au_q = 0;
  if (args.audit_list != nullptr) {
    const unsigned u = (unsigned)bid * 8u + (unsigned)wave;
    if ((int)(u % (unsigned)args.audit_div) == args.audit_phase) {
      const unsigned h = audit_hash(u, args.audit_seed);
      au_slot = (int)(u / (unsigned)args.audit_div);
      au_rw = (int)(h & 15u);
      au_q = (int)((h >> 4) & 7u);
      au_lane = (int)((h >> 7) & 63u);
    }
  }
  if ((int64_t)row_tile * 256 >= args.n) {           // (whole workgroup: the grid is padded to a multiple of 8 row tiles)
    if (au_slot >= 0 && lane == 0) args.audit_list[au_slot] = -1;
    return;
  }
  const int ktiles = args.ktiles;
  const int stages = 2 * ktiles, lasts = stages - 1;
  const char* img = reinterpret_cast<const char*>(args.image) + (size_t)cb * ktiles * 32768;
  const int64_t blk_row0 = (int64_t)row_tile * 256;
  const int64_t row0 = blk_row0 + wave * kWaveRows;
  const char* xblk = reinterpret_cast<const char*>(args.X + blk_row0 * args.ldx);

  // DMA offsets.  x is staged in FULL 128-byte lines: piece j (0..3) of a wave = rows 8j..8j+7 of its 32; lane
  // l = (r = l>>3, q = l&7) fetches 16-byte chunk q ^ r ^ (j&1) of row 8j + r, so an 8-lane group covers one whole
  // line (in permuted order) and the read-back (each lane: its row's two chunks of the k-tile quarter it feeds) is
  // conflict-free for ds_read_b128's 16-lane groups.
  unsigned poff[kPP], xfo[2 * RT], xrd[RT][2];
#pragma unroll
  for (int q = 0; q < kPP; ++q) poff[q] = (unsigned)(((W * q + wave) * 64 + lane) * 16);
  {
    const int r8 = lane >> 3, q8 = lane & 7;
#pragma unroll
    for (int j = 0; j < 2 * RT; ++j) {
      const int64_t r = row0 + 8 * j + r8;
      const int64_t rl = (r < args.n ? r : args.n - 1) - blk_row0;   // clamp: loads stay in bounds, stores are masked
      xfo[j] = (unsigned)((rl * args.ldx + 4 * (q8 ^ r8 ^ (j & 1))) * 4);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int R = 16 * rt + r16, j = R >> 3, r = R & 7;       // this lane's row of row tile rt: chunks 2g, 2g+1
#pragma unroll
      for (int c = 0; c < 2; ++c) xrd[rt][c] = (unsigned)(j * 1024 + (r * 8 + ((2 * g + c) ^ r ^ (j & 1))) * 16);
    }
  }

  f32x4 acc[RT][16];
  float ss[RT], sm[RT], amax[RT];      // ||x_hi||^2, ||x_mid||^2 (both from the bf16 pieces the MFMAs consume), max |x|
  auto zero_tile_state = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < 16; ++ct) {
        acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // pin the zeroing HERE: the MFMAs below are inline asm, so hipcc pads no hazard for them - a v_accvgpr_write
        // rematerialised right in front of the first accumulation would be read too early
        asm volatile("" : "+a"(acc[rt][ct]));
      }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { ss[rt] = 0.f; sm[rt] = 0.f; amax[rt] = 0.f; }
  };
  unsigned long long t_shader = 0, t_real = 0;
  if (args.clock_probe != nullptr) {
    t_shader = __builtin_amdgcn_s_memtime();
    t_real = __builtin_amdgcn_s_memrealtime();
  }

  struct Dma { const char* pg; const char* xg; float* pdst; float* xdst; int j0; };
  const int last_valid_chunks = PARTIAL ? (args.dim - (ktiles - 1) * kKTile) / 4 : 8;   // 16-byte chunks of a row in the last k-tile
  auto plan = [&](int s) {          // what stage s issues: fragments of stage s+2, x pieces 4(s&1).. of tile (s>>1)+2
    Dma f;
    const int ns = s + 2, c = ns < lasts ? ns : lasts;
    const int nt = (s >> 1) + 2, t = nt < ktiles ? nt : ktiles - 1;
    f.pg = img + (size_t)c * 16384;
    f.xg = xblk + (size_t)t * (kKTile * 4);
    f.pdst = lds + (ns % 3) * kPHalf + wave * kFragFloats;
    f.j0 = kXPS * (s & 1);
    f.xdst = lds + 3 * kPHalf + (nt % 3) * kXTile + wave * kXWave + f.j0 * kFragFloats;
    return f;
  };
  auto issue = [&](const Dma& f, int d) {
    if constexpr (!PARTIAL) {
      if (d < kPP)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.pg + poff[d]), (LDS_AS void*)(f.pdst + W * d * kFragFloats),
                                         16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.xg + (f.j0 ? xfo[kXPS + d - kPP] : xfo[d - kPP])),
                                         (LDS_AS void*)(f.xdst + (d - kPP) * kFragFloats), 16, 0, LSHRS_X_AUX);
    } else if (d < kPP) {
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.pg + poff[d]), (LDS_AS void*)(f.pdst + W * d * kFragFloats),
                                       16, 0, 0);
    } else {
      // branch-free (a branch here changes where hipcc joins the accumulator tiles around the inline-asm MFMAs): in the last
      // k-tile a chunk past the row's end is fetched from the tile's first chunk instead; this lane's chunk of the line, as in xfo
      const unsigned off = f.j0 ? xfo[kXPS + d - kPP] : xfo[d - kPP];
      const int lim = f.xg == xblk + (size_t)(ktiles - 1) * (kKTile * 4) ? last_valid_chunks : 8;
      const unsigned chunk = (unsigned)((lane & 7) ^ (lane >> 3) ^ ((f.j0 + d - kPP) & 1));
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.xg + (off - ((int)chunk >= lim ? 16u * chunk : 0u))),
                                       (LDS_AS void*)(f.xdst + (d - kPP) * kFragFloats), 16, 0, LSHRS_X_AUX);
    }
  };
  f32x4 xr[RT][2];                           // raw f32 x of one k-tile: [row tile][chunk]
  auto read_x = [&](int t) {
    const char* xt = reinterpret_cast<const char*>(lds + 3 * kPHalf + (t % 3) * kXTile + wave * kXWave);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < 2; ++c) xr[rt][c] = *reinterpret_cast<const f32x4*>(xt + xrd[rt][c]);
    if constexpr (PARTIAL) {                // chunks 2 g, 2 g + 1 of the last k-tile: past the row's end they read as zero (selects, no branch)
      const int lim = t >= ktiles - 1 ? last_valid_chunks : 8;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bool gone = 2 * g + c >= lim;
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[rt][c][e] = gone ? 0.f : xr[rt][c][e];
        }
    }
  };
  float r0 = 0.f, r1 = 0.f;
  auto split_step = [&](int q, Bf16Pairs (&hi)[RT], Bf16Pairs (&mid)[RT]) {   // slice q (0..kSlices-1) of one k-tile's split
    const int pair = q / 3, step = q % 3, rt = pair >> 2, pr = pair & 3, c = pr >> 1, e = 2 * (pr & 1);
    const float v0 = xr[rt][c][e], v1 = xr[rt][c][e + 1];
    if (step == 0) {
      const bf16x2 hp = bf16x2{(__bf16)v0, (__bf16)v1};
      hi[rt].p[pr] = hp;
      r0 = v0 - (float)hp[0];
      r1 = v1 - (float)hp[1];
    } else if (step == 1) {
      mid[rt].p[pr] = bf16x2{(__bf16)r0, (__bf16)r1};
    } else {
      ss[rt] = __builtin_amdgcn_fdot2_f32_bf16(hi[rt].p[pr], hi[rt].p[pr], ss[rt], false);
#ifndef LSHRS_AB_NO_XMID_NORM          // (A/B builds only, tools/ab_build.py: what ||x_mid||^2 costs stage 1; keys are wrong without it)
      sm[rt] = __builtin_amdgcn_fdot2_f32_bf16(mid[rt].p[pr], mid[rt].p[pr], sm[rt], false);
#endif
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax[rt]) : "v"(v0), "v"(v1));
    }
  };
  // Fragments travel in EIGHTHS of a stage: 2 column tiles x {hi, mid} = 4 ds_read_b128 = 16 VGPRs, two buffers.
  // (Quarters, as in sig_kernel, push this kernel over 256 VGPRs next to its 256 accumulator AGPRs: hipcc then
  // shuffles accumulators through v_accvgpr moves inside the loop.)
  auto read_eighth = [&](const float* base, int e, f32x4 (&f)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f[j][0] = *reinterpret_cast<const f32x4*>(base + (((2 * e + j) * 2 + 0) * 64 + lane) * 4);
      f[j][1] = *reinterpret_cast<const f32x4*>(base + (((2 * e + j) * 2 + 1) * 64 + lane) * 4);
    }
  };
  // MFMA k (0..11) of an eighth: term k / 4 of xh*ph + xh*pm + xm*ph, column tile ct0 + (k / 2) % 2, row tile k % 2: the
  // four accumulator tiles of the eighth take turns, so two MFMAs on the same tile are four instructions (64 cycles)
  // apart.  (Two apart - tile order (j, term, rt) - the result of a 4-pass MFMA is not back in time and hipcc pads every
  // other MFMA with an s_nop: 33 per k-tile and wave.)  Every tile still sees its terms in the order 0, 1, 2.
  auto mfma_one = [&](int ct0, int k, const f32x4 (&f)[2][2], const Bf16Pairs (&hi)[RT], const Bf16Pairs (&mid)[RT]) {
    const int term = k / (2 * RT), j = (k / RT) % 2, rt = k % RT;
    const bf16x8 a = __builtin_bit_cast(bf16x8, term == 2 ? mid[rt] : hi[rt]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, f[j][term == 1 ? 1 : 0]);
    // Inline asm pins the accumulator to AGPRs and to in-place accumulation: left to the builtin, hipcc renames
    // accumulator tiles between MFMAs (vDst != SrcC) and parks some in VGPRs, i.e. hundreds of v_accvgpr moves and
    // s_nops per loop body.  Dependent MFMAs are four instructions (64 cycles) apart, beyond the 4-pass hazard window.
#ifdef LSHRS_T16_BUILTIN
    acc[rt][ct0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[rt][ct0 + j], 0, 0, 0);
#else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[rt][ct0 + j]) : "v"(a), "v"(b));
#endif
  };

  f32x4 fa[2][2], fb[2][2];
  Bf16Pairs hs0[RT], ms0[RT], hs1[RT], ms1[RT];   // bf16 x of the k-tiles, ping-pong by tile parity

  // One k-tile = two stages (column halves ch), one stage = four eighths E0..E3 of 24 MFMAs.  Every eighth's MFMAs
  // run while the next eighth's fragments are read; the last eighth of a stage is consumed after the barrier:
  //   barrier(s) | read E0(s) | MFMA E3(s-1) | read E1(s) [+ raw x of the next tile, ch = 1] | MFMA E0(s) |
  //              | read E2(s) | MFMA E1(s) + DMA | read E3(s) | MFMA E2(s) + DMA | barrier(s+1)
  // The next tile's x is read in the second stage of a tile and split (48 slices) under that stage's last two
  // eighths and the first eighth after the tile boundary: it must be complete before E0 of the new tile.
  auto stage = [&](int s, const int ch, const bool first, const Bf16Pairs (&hc)[RT], const Bf16Pairs (&mc)[RT],
                   const Bf16Pairs (&hp)[RT], const Bf16Pairs (&mp)[RT], Bf16Pairs (&hn)[RT], Bf16Pairs (&mn)[RT]) {
    // hc/mc: this stage's tile; hp/mp: the tile E3(s-1) belongs to; hn/mn: where the split in flight writes
    const float* st = lds + (s % 3) * kPHalf;
    read_eighth(st, 0, fa);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kE; ++k) {
      if (!first) mfma_one(8 * (1 - ch) + 6, k, fb, hp, mp);        // E3 of the previous stage (the other column half)
      if (ch == 0) {                                                  // the last third of this tile's split (2 slices per 3 MFMAs)
        if (k % 3 != 2) split_step(2 * kSlices / 3 + (k / 3) * 2 + k % 3, hn, mn);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<kXPS>();                                               // own fragments of stage s+1 and every older x piece
    read_eighth(st, 1, fb);
    if (ch == 1) read_x((s >> 1) + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kE; ++k) {
      mfma_one(8 * ch + 0, k, fa, hc, mc);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    read_eighth(st, 2, fa);
    __builtin_amdgcn_sched_barrier(0);
    const Dma f = plan(s);
#pragma unroll
    for (int k = 0; k < kE; ++k) {
      mfma_one(8 * ch + 2, k, fb, hc, mc);
      if (ch == 1 && k % 3 != 2) split_step((k / 3) * 2 + k % 3, hn, mn);          // first third of the next tile's split
      if (k % 6 == 0) issue(f, k / 6);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    read_eighth(st, 3, fb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kE; ++k) {
      mfma_one(8 * ch + 4, k, fa, hc, mc);
      if (ch == 1 && k % 3 != 2) split_step(kSlices / 3 + (k / 3) * 2 + k % 3, hn, mn);   // second third
      if (k % 6 == 0) issue(f, kE / 6 + k / 6);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  // tile t with its sets (hc, mc); the previous tile's (hp, mp) double as the target of the next tile's split
  auto tile = [&](int t, const bool first, Bf16Pairs (&hc)[RT], Bf16Pairs (&mc)[RT], Bf16Pairs (&hp)[RT], Bf16Pairs (&mp)[RT]) {
    stage(2 * t, 0, first, hc, mc, hp, mp, hc, mc);       // ch 0: E3(s-1) is the previous tile's; the split in flight is this tile's
    stage(2 * t + 1, 1, false, hc, mc, hc, mc, hp, mp);   // ch 1: E3(s-1) is this tile's; the next tile's split starts (into the other set)
  };

  auto issue_prologue = [&]() {     // the first two stages' fragments and the first two x tiles of the row tile entered
    const Dma a0 = plan(-4), a1 = plan(-3), b0 = plan(-2), b1 = plan(-1);
#pragma unroll
    for (int d = 0; d < kPP; ++d) issue(b0, d);                                  // fragments of stage 0
#pragma unroll
    for (int d = kPP; d < kPP + kXPS; ++d) { issue(a0, d); issue(a1, d); }       // x tile 0
#pragma unroll
    for (int d = 0; d < kPP; ++d) issue(b1, d);                                  // fragments of stage 1
#pragma unroll
    for (int d = kPP; d < kPP + kXPS; ++d) { issue(b0, d); issue(b1, d); }       // x tile 1
  };
  // Static priority for the second-dispatched half of the workgroup: of the two waves of a SIMD the younger one loses the
  // VALU arbitration (priority, then age) on every stage; one s_setprio for that half, no flips (MI355X_MICROARCH.md, "Two
  // waves per SIMD", item 4).  Same box, interleaved, four pairs: +0.2 .. +0.9 % (profiles/r03_static_prio_ab.log).
#ifndef LSHRS_AB_NO_STATIC_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  issue_prologue();
  zero_tile_state();
  wait_vmcnt<kPP + 2 * kXPS>();
  __builtin_amdgcn_s_barrier();
  read_x(0);
#pragma unroll
  for (int q = 0; q < 2 * kSlices / 3; ++q) split_step(q, hs0, ms0);   // tile 0 only: the last third rides in stage 0 as for every tile
  tile(0, true, hs0, ms0, hs1, ms1);
  int t = 1;
  for (; t + 1 < ktiles; t += 2) {
    tile(t, false, hs1, ms1, hs0, ms0);
    tile(t + 1, false, hs0, ms0, hs1, ms1);
  }
  if (t < ktiles) {                                                      // even number of k-tiles: one more, then drain with its set
    tile(t, false, hs1, ms1, hs0, ms0);
#pragma unroll
    for (int k = 0; k < kE; ++k) { mfma_one(14, k, fb, hs1, ms1); asm volatile("s_nop 7\n\ts_nop 4"); }
  } else {
#pragma unroll
    for (int k = 0; k < kE; ++k) { mfma_one(14, k, fb, hs0, ms0); asm volatile("s_nop 7\n\ts_nop 4"); }
  }
  // (the wait states after every MFMA of the drain: where the two branches join hipcc may copy accumulator tiles, and
  //  it does not know that the asm in front of such a copy is an MFMA whose result takes passes to arrive)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped prefetches past the last stage must land before exit
  __builtin_amdgcn_s_barrier();
  // Landing point of the accumulators.  hipcc does not know that the asm statements above are MFMAs whose results
  // take passes to arrive: without this it hoists the epilogue's v_accvgpr_reads to a few instructions behind the
  // last MFMA (observed: one register of one tile read before its final accumulation).  Volatile asms keep their
  // order, and every read below depends on the empty asm that follows the wait states.
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) asm volatile("" : "+a"(acc[rt][ct]));

  if (args.clock_probe != nullptr && tid == 0) {
    const unsigned long long slot = (unsigned long long)blockIdx.x;
    args.clock_probe[2 * slot] = __builtin_amdgcn_s_memtime() - t_shader;
    args.clock_probe[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - t_real;
  }

  // The ring is free (every wave is past the barrier above, every prefetch has landed): the epilogue stages the
  // workgroup's flagged projections in it - list entries and their stage-1 values - and appends them to the global
  // list with ONE atomic per workgroup.  (One atomic per flagged projection on the single global counter serialises:
  // at a 870-unit window - 300 k flagged projections per 1M rows - it tripled the kernel's time.)
  // Everything the epilogue derives from the lane index is loop-invariant: left alone, hipcc computes it in front of
  // the main loop and carries (or spills) it across.  These copies are opaque: the epilogue's addressing starts here.
  int r16e = r16, ge = g, lanee = lane;
  asm volatile("" : "+v"(r16e), "+v"(ge), "+v"(lanee));
  int64_t* l_list = reinterpret_cast<int64_t*>(lds);
  float* l_y = lds + 2 * kS1ListCap;
  int* l_count = reinterpret_cast<int*>(lds + kRingFloats + 512);   // [0] staged + overflowed entries, [1] global base
  if (tid == 0) l_count[0] = 0;

  // ---- row statistics -> the two factors of the stage-1 window per row ------------------------------------------------
  // ||x_hi|| and ||x_mid|| are sums over the very bf16 values the matrix instructions consumed (f32 accumulation: + 0.1 %,
  // which also covers what separates ||x_hi|| + ||x_mid|| from ||x||).  A row whose largest |x| is outside
  // [2^-32, 2^32] leaves the range in which the squares and the split neither underflow nor overflow: all of its
  // projections are re-evaluated (NOT(|y| > +inf) holds for every y).  A true zero row gives y = 0 in both passes.
  float* wnd_lds = lds + kRingFloats + wave * kWaveRows;
  float* wnb_lds = lds + kRingFloats + 256 + wave * kWaveRows;
  // the window coefficients of this column block, staged once (behind the list stage, which owns the first 3 x kS1ListCap
  // floats of the ring): with the proven window the exact test below runs on a third of the 32-column words, and a
  // global load in front of each of its compares is latency two waves per SIMD cannot hide
  float* coef_lds = lds + 3 * kS1ListCap;
  static_assert(3 * kS1ListCap + 512 <= kRingFloats, "coefficients behind the list stage");
  coef_lds[tid] = tid < 256 ? args.wa[cb * 256 + tid] : args.wb[cb * 256 + tid - 256];
  // compact column blocks (sig_compact): the padded id of every column of this block, and room for the block's sign words
  int* padcol_lds = reinterpret_cast<int*>(lds + 3 * kS1ListCap + 512);
  uint32_t* cw_lds = reinterpret_cast<uint32_t*>(lds + 3 * kS1ListCap + 768);
  static_assert(3 * kS1ListCap + 768 + 256 * 8 <= kRingFloats, "compact tables behind the coefficients");
  if (COMPACT && tid < 256) padcol_lds[tid] = args.padcol[cb * 256 + tid];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float s2 = ss[rt] + __shfl_xor(ss[rt], 16);
    s2 += __shfl_xor(s2, 32);
    float m2 = sm[rt] + __shfl_xor(sm[rt], 16);
    m2 += __shfl_xor(m2, 32);
    float am = __builtin_fmaxf(amax[rt], __shfl_xor(amax[rt], 16));
    am = __builtin_fmaxf(am, __shfl_xor(am, 32));
    const int64_t myrow = row0 + 16 * rt + r16e;
    if (ge == 0) {
      float window = sqrtf(s2) * args.tau * 1.001f;
      if (am != 0.f && !(am >= 0x1p-32f && am <= 0x1p32f)) window = __builtin_inff();
      wnd_lds[16 * rt + r16e] = window;
      const float wb_ = sqrtf(m2) * args.tau_b * 1.001f;
      wnb_lds[16 * rt + r16e] = wb_ < __builtin_inff() ? wb_ : 0.f;      // (NaN / Inf rows: the first factor decides)
      if (cb == 0 && args.row_flags != nullptr && myrow < args.n) {
        const bool has_nan = s2 != s2;
        const bool zero = (am <= 1e-8f) && !has_nan;
        args.row_flags[myrow] = (uint8_t)((zero ? 1 : 0) | (has_nan ? 2 : 0));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();

  // ---- sign bits.  One v_cmp per accumulator register = 4 rows (g') x 16 columns: its low 32 bits are rows g' = 0, 1,
  // its high 32 bits rows g' = 2, 3 of the tile.  Lane L owns the ROW PAIR p = L / 4 = (rtl, g'-pair, reg) - rows
  // 16 rtl + 8 g'pair + reg and + 4 - and the 32-column words 2 (L % 4), + 1: the ballot halves of the even column
  // tile land in A[], of the odd one in B[] (deposit_positive: v_cmp, the two wait states a VALU-written SGPR needs,
  // two v_writelane), and two VALU ops per word merge the 16-bit halves.
  const float amax_cb = args.wamax[cb], bmax_cb = args.wbmax[cb];
  {
    uint32_t A[2] = {0u, 0u}, B[2] = {0u, 0u};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const f32x4 wnd = *reinterpret_cast<const f32x4*>(wnd_lds + 16 * rt + 4 * ge);   // rows 16 rt + 4 g + 0..3
      const f32x4 wnb = *reinterpret_cast<const f32x4*>(wnb_lds + 16 * rt + 4 * ge);
      // per-lane screen: the largest window of this lane's four rows (a non-finite window - NaN or Inf in the row, or
      // a magnitude outside the guarded range - makes it +inf: everything goes to the exact test)
      float tsmax = __builtin_fmaxf(__builtin_fmaxf(wnd[0], wnd[1]), __builtin_fmaxf(wnd[2], wnd[3]));
      if (!(wnd[0] < __builtin_inff()) || !(wnd[1] < __builtin_inff()) || !(wnd[2] < __builtin_inff()) ||
          !(wnd[3] < __builtin_inff()))
        tsmax = __builtin_inff();
      tsmax = tsmax * amax_cb +
              __builtin_fmaxf(__builtin_fmaxf(wnb[0], wnb[1]), __builtin_fmaxf(wnb[2], wnb[3])) * bmax_cb;
      tsmax = tsmax > 0.f ? tsmax : -1.f;               // all four rows zero: nothing to re-evaluate
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        float m = __builtin_inff();                     // min |y| over the 2 tiles x 4 registers of this word (NaN dropped:
                                                        // a NaN y only comes from a row whose window is non-finite)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float y0 = acc[rt][2 * w][reg], y1 = acc[rt][2 * w + 1][reg];
          const int p0 = 8 * rt + reg * 2;              // pair (rt, reg, g'pair = 0); g'pair = 1 is p0 + 1
          deposit_positive(A[w & 1], y0, 4 * p0 + (w >> 1), 4 * (p0 + 1) + (w >> 1));
          deposit_positive(B[w & 1], y1, 4 * p0 + (w >> 1), 4 * (p0 + 1) + (w >> 1));
          asm("v_min3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(y0), "v"(y1));
        }
        const bool aud = au_rw == 8 * rt + w;                   // (wave-uniform: this word holds the wave's audit sample)
        if (__builtin_amdgcn_ballot_w64(!(m > tsmax)) != 0 || aud) {   // wave-uniform: the exact per-element test
          // With the proven window this runs on a quarter of the words: first the eight comparisons, branch-free, into a
          // mask; only the lane that holds a flagged projection (one, seldom two of the wave) enters the append.
          unsigned hits = 0u;
          float ys[8], thrs[8];
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int ct = 2 * w + half;
            const float pa = coef_lds[16 * ct + r16e], pb = coef_lds[256 + 16 * ct + r16e];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              float thr = wnd[reg] * pa + wnb[reg] * pb;
              thr = thr > 0.f ? thr : -1.f;                               // zero row / zero-padded column: y is exactly 0
              ys[4 * half + reg] = acc[rt][ct][reg];
              thrs[4 * half + reg] = thr;
              hits |= (!(__builtin_fabsf(ys[4 * half + reg]) > thr) ? 1u : 0u) << (4 * half + reg);
            }
          }
          if (aud && lanee == au_lane) {
            // the sample: value au_q of this lane - left for stage 2 with the window it has just been compared with, unless
            // it is flagged anyway (then stage 2 decides it), sits in a padding column or past the last row
            float yq = ys[0], tq = thrs[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
              yq = au_q == q ? ys[q] : yq;
              tq = au_q == q ? thrs[q] : tq;
            }
            const int64_t grow = row0 + 16 * rt + 4 * ge + (au_q & 3);
            const int ct = 2 * w + (au_q >> 2);
            const int colid = COMPACT ? padcol_lds[16 * ct + r16e] : cb * 256 + 16 * ct + r16e;
            const bool keep = ((hits >> au_q) & 1u) == 0u && grow < args.n && colid >= 0 && colid < args.row_bytes * 8 &&
                              tq < __builtin_inff();
            args.audit_list[au_slot] = keep ? ((grow << 21) | (int64_t)colid) : (int64_t)-1;
            args.audit_vals[2 * au_slot] = yq;
            args.audit_vals[2 * au_slot + 1] = tq;
          }
          if (hits != 0u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int reg = q & 3, ct = 2 * w + (q >> 2);
              const int64_t grow = row0 + 16 * rt + 4 * ge + reg;
              const int colid = COMPACT ? padcol_lds[16 * ct + r16e] : cb * 256 + 16 * ct + r16e;
              if (((hits >> q) & 1u) != 0u && grow < args.n && (!COMPACT || colid >= 0)) {
                const int64_t entry = (grow << 21) | (int64_t)colid;
                // the stage-1 value travels with the entry: stage 2 measures |y1 - y_BLAS| on every flagged projection
                // (rows flagged wholesale carry no usable y1: NaN, skipped by that statistic)
                const float ykeep = wnd[reg] < __builtin_inff() ? ys[q] : __builtin_nanf("");
                const int pos = atomicAdd(l_count, 1);                    // LDS atomic
                if (pos < kS1ListCap) {
                  l_list[pos] = entry;
                  l_y[pos] = ykeep;
                } else {                                                  // LDS stage full (rows flagged wholesale): straight out
                  const int slot = atomicAdd(args.tie_count, 1);
                  if (slot < args.tie_cap) {
                    args.tie_list[slot] = entry;
                    if (args.flag_y != nullptr) args.flag_y[slot] = ykeep;
                  }
                }
              }
            }
          }
        }
      }
    }
    // lane L: pair p = L / 4 -> rows lo / lo + 4, words 2 (L % 4), + 1
    const int pr = lanee >> 2, wq = 2 * (lanee & 3);
    const int rlo = 16 * (pr >> 3) + 8 * (pr & 1) + ((pr >> 1) & 3);
    const uint32_t wlo[2] = {(A[0] & 0xFFFFu) | (B[0] << 16), (A[1] & 0xFFFFu) | (B[1] << 16)};
    const uint32_t whi[2] = {(A[0] >> 16) | (B[0] & 0xFFFF0000u), (A[1] >> 16) | (B[1] & 0xFFFF0000u)};
    const int byte0 = (cb * 8 + wq) * 4;
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int64_t grow = row0 + rlo + 4 * hl;
      if (COMPACT) {                  // the block's sign string of this row: to LDS, the key bytes are cut from it below
        uint32_t* dstw = cw_lds + (wave * kWaveRows + rlo + 4 * hl) * 8 + wq;
        dstw[0] = hl ? whi[0] : wlo[0];
        dstw[1] = hl ? whi[1] : wlo[1];
      } else if (grow < args.n) {
        uint8_t* dst = args.keys + grow * (int64_t)args.row_bytes + byte0;
        const uint32_t w0 = hl ? whi[0] : wlo[0], w1 = hl ? whi[1] : wlo[1];
        if (args.vec_store && byte0 + 8 <= args.row_bytes) {
          *reinterpret_cast<u32x2*>(dst) = u32x2{w0, w1};
        } else {
#pragma unroll
          for (int bsel = 0; bsel < 4; ++bsel) {
            if (byte0 + bsel < args.row_bytes) dst[bsel] = (uint8_t)(w0 >> (8 * bsel));
            if (byte0 + 4 + bsel < args.row_bytes) dst[4 + bsel] = (uint8_t)(w1 >> (8 * bsel));
          }
        }
      }
    }
  }

  // ---- the workgroup's flagged projections: one global append ------------------------------------------------------
  __syncthreads();
  if (COMPACT) {
    // key byte o of this block = bits [src, src + 8) of the row's 256-bit sign string, masked to the band's live rows
    const int bands_here = args.num_bands - cb * args.bpb < args.bpb ? args.num_bands - cb * args.bpb : args.bpb;
    const int nby = bands_here * args.band_bytes;
    const int byte_base = cb * args.bpb * args.band_bytes;
    const int* tab = args.bytetab + cb * 512;
    for (int idx = tid; idx < 256 * nby; idx += 64 * W) {
      const int rl = idx / nby, o = idx - rl * nby;
      const int64_t grow = blk_row0 + rl;
      if (grow < args.n) {
        const int src = tab[2 * o], w = src >> 5;
        const uint32_t lo = cw_lds[rl * 8 + w], hi = cw_lds[rl * 8 + (w < 7 ? w + 1 : 7)];
        const uint32_t v = (uint32_t)((((uint64_t)hi << 32) | lo) >> (src & 31)) & (uint32_t)tab[2 * o + 1];
        args.keys[grow * (int64_t)args.row_bytes + byte_base + o] = (uint8_t)v;
      }
    }
  }
  const int staged = l_count[0] < kS1ListCap ? l_count[0] : kS1ListCap;
  if (staged > 0) {                                     // (workgroup-uniform)
    if (tid == 0) l_count[1] = atomicAdd(args.tie_count, staged);
    __syncthreads();
    const int base = l_count[1];
    for (int e = tid; e < staged; e += 64 * W) {
      const int slot = base + e;
      if (slot < args.tie_cap) {
        args.tie_list[slot] = l_list[e];
        if (args.flag_y != nullptr) args.flag_y[slot] = l_y[e];
      }
    }
  }
  if (args.clock_probe != nullptr && tid == 0) {
    const unsigned long long slot = (unsigned long long)gridDim.x + (unsigned long long)blockIdx.x;
    args.clock_probe[2 * slot] = __builtin_amdgcn_s_memtime() - t_shader;
    args.clock_probe[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - t_real;
  }
}

// ------------------------------------------------------------------------------------------
// Stage 1 of the split pass for SHORT vectors and narrow hashers (dim <= 256 and nct x kt <= 64, at most 256 key columns: BASELINE config
// 1's 16 x 4 x 128, the reference's docstring layout 20 x 6 x 128, num_perm = 128 at 128-d): the whole bf16 hi / mid
// fragment image stays RESIDENT in LDS (KT k-tiles x NCT 16-column tiles x 2 KiB: 64 KiB at 128 x 128) and every wave
// walks 32-row tiles of x on its own - no ring, no barrier after the prologue, nothing staged per tile but x itself.
// sig16_kernel spends a prologue and an epilogue per 256 rows around four k-tiles of such a shape, and the exact-f32 kernel is
// bound by the f32 matrix rate there (1 M x 128 x 128 columns: 0.32 ms = 0.7 of that roof, 0.10 of the HBM roof).
//   wave   = RT (two, or one where the registers ask for it: res_rt) 16-row tiles x NCT column tiles, one of eight in a
//            persistent workgroup (one workgroup per CU, two waves per SIMD within 256 registers each: the second wave is
//            what hides a wave's LDS round trips); tile i of the batch goes to wave i mod (8 x workgroups);
//   x      straight from HBM to registers: lane (r, g) owns elements 32 t + 8 g .. + 7 of row r - an operand of
//            v_mfma_f32_16x16x32_bf16 as it comes (the B operand: the accumulators are P X^T, lane (r, g) ends with columns
//            4 g .. + 3 of every column tile for ITS row r); the registers of k-tile t are refilled with the NEXT tile's
//            elements as soon as k-tile t has been split, so a whole tile of loads is in flight under the matrix work;
//   order  per k-tile x_hi p_hi, x_hi p_mid, x_mid p_hi on every accumulator, k-tiles ascending: the accumulation
//            lshrs_split_stage1_model states and the proven window (lshrs_sig_set_window) is derived for - same
//            coefficients, same stage 2;
//   keys   the column block is COMPACT (sig_compact's layout: the bands' rows side by side, no padding columns) and packed in
//            the order that makes a lane's values consecutive bits of its row's sign string (res_colmap): one v_alignbit per
//            value; the strings go through the wave's own LDS patch and leave as key bytes through the byte table;
//   list   flagged projections are staged per wave in LDS and leave with one global atomic per 64 .. 128 entries.
// ------------------------------------------------------------------------------------------
// waves per workgroup (one workgroup per CU): two per SIMD (<= 256 registers each), three where one row tile over <= 32
// (column tile, k-tile) pairs fits 168 registers - the other waves are what hides a wave's LDS and memory round trips
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

template <int NCT, int KT>
__global__ __launch_bounds__(64 * res_waves(NCT, KT), 1) void sig16r_kernel(const SigArgs args) {
  constexpr int RT = res_rt(NCT, KT), NW = NCT / 2, kRows = 16 * RT, kResWaves = res_waves(NCT, KT), kResGroups = (NCT + 7) / 8;
  constexpr int kImgFloats = KT * NCT * 512;
  constexpr int kResWaveFloats = res_wave_floats(RT);
  __shared__ __attribute__((aligned(16))) float lds[res_lds_floats<NCT, KT>()];
  struct Bf16Pairs { bf16x2 p[4]; };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, g = lane >> 4;
  float* coef_lds = lds + kImgFloats;                                  // wa[256] | wb[256] of the (compact) block
  int* padcol_lds = reinterpret_cast<int*>(lds + kImgFloats + 512);
  int* tab_lds = reinterpret_cast<int*>(lds + kImgFloats + 768);       // byte table: (source bit, mask) per key byte
  float* mine = lds + kImgFloats + 1280 + wave * kResWaveFloats;       // this wave's patch
  uint32_t* cw_lds = reinterpret_cast<uint32_t*>(mine + 2 * kRows);    // [rows of the tile][8 words + one that is only ever read]
  int64_t* l_list = reinterpret_cast<int64_t*>(mine + 11 * kRows);
  float* l_y = mine + 11 * kRows + 2 * kResListCap;
  int* l_count = reinterpret_cast<int*>(mine + 11 * kRows + 3 * kResListCap);

  // ---- prologue: the image (L2 -> LDS, [kt][ct < NCT][part][lane] x 16 B) and the block's tables --------------------
  {
    const f32x4* img = reinterpret_cast<const f32x4*>(args.image);     // global: [kt][16 ct][part][lane]
#ifdef LSHRS_AB_RES_COPY_PROLOGUE
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int c = tid; c < KT * NCT * 128; c += 64 * kResWaves) {
      const int l = c & 127, ct = (c >> 7) % NCT, kt = (c >> 7) / NCT;
      dst[c] = img[(kt * 16 + ct) * 128 + l];
    }
#else
    // LDS-DMA, every piece of the image in flight at once (a copy through registers is a chain of round trips: -5 us a launch)
    for (int c0 = wave * 64; c0 < KT * NCT * 128; c0 += 64 * kResWaves) {          // (uniform: 64 chunks of 16 B per wave and step)
      const int c = c0 + lane, l = c & 127, ct = (c >> 7) % NCT, kt = (c >> 7) / NCT;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(img + (kt * 16 + ct) * 128 + l), (LDS_AS void*)(lds + 4 * c0), 16, 0, 0);
    }
#endif
    if (tid < 256) {
      coef_lds[tid] = args.wa[tid];
      coef_lds[256 + tid] = args.wb[tid];
      padcol_lds[tid] = args.padcol[tid];
      tab_lds[tid] = args.bytetab[2 * tid] | (args.bytetab[2 * tid + 1] << 16);      // source bit | mask of the live bits
    }
    if (lane == 0) l_count[0] = 0;
#ifndef LSHRS_AB_RES_COPY_PROLOGUE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();

  const int64_t tiles = (args.n + kRows - 1) / kRows;
  const int64_t stride = (int64_t)gridDim.x * kResWaves;
  const int dim = args.dim;
  const float amax_cb = args.wamax[0], bmax_cb = args.wbmax[0];
  const int nby = args.row_bytes;

  // x of one 32-row tile: [row tile][k-tile][chunk of four].  Every load is unconditional (a predicated load is a branch, and
  // a wait for everything in flight behind it): a chunk past the row's end (dim % 4 == 0) is fetched from the row's last
  // chunk instead and zeroed when its k-tile is split.
  f32x4 xr[RT][KT][2];
  int koff[KT][2];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) koff[t][c] = 32 * t + 8 * g + 4 * c < dim ? 32 * t + 8 * g + 4 * c : dim - 4;
  auto load_x = [&](int64_t tile, int t) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int64_t row = tile * kRows + 16 * rt + r16;
      row = row < args.n ? row : args.n - 1;                           // clamp: loads stay in bounds, stores are masked
#ifdef LSHRS_AB_RES_L2ROWS        // (A/B builds only: every tile reads the batch's first rows - from L2: the kernel without the HBM stream; wrong keys by design)
      row &= (LSHRS_AB_RES_L2ROWS - 1);
#endif
      const float* xp = args.X + row * args.ldx;
#pragma unroll
      for (int c = 0; c < 2; ++c) xr[rt][t][c] = *reinterpret_cast<const f32x4*>(xp + koff[t][c]);
    }
  };
  int64_t tile = (int64_t)blockIdx.x * kResWaves + wave;
  if (tile < tiles) {
#pragma unroll
    for (int t = 0; t < KT; ++t) load_x(tile, t);
  }

#ifdef LSHRS_AB_RES_PROBE        // (A/B builds only, tools/res_phase_probe.py: shader cycles a wave spends in its k-loops and in its epilogues)
  unsigned long long pr_main = 0, pr_epi = 0, pr_tiles = 0;
  const unsigned long long pr_t00 = __builtin_amdgcn_s_memtime(), pr_r00 = __builtin_amdgcn_s_memrealtime();
#endif
  for (; tile < tiles; tile += stride) {
#ifdef LSHRS_AB_RES_PROBE
    const unsigned long long pr_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int64_t row0 = tile * kRows;
    const int64_t next = tile + stride < tiles ? tile + stride : tile;  // (the last tile re-fetches itself: unused, in bounds)
    f32x4 acc[RT][NCT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss[RT] = {}, sm[RT] = {}, amax[RT] = {};
#ifndef LSHRS_AB_RES_NO_PRIO
    // the k-loop is where a wave issues the loads of its NEXT tile: it goes ahead of the waves that are in their epilogues,
    // so that the memory pipeline is fed on time (the kernel's floor is the x stream: profiles/r04_resident_attribution.log)
    __builtin_amdgcn_s_setprio(2);
#endif

#ifdef LSHRS_AB_RES_NO_MAIN   // (A/B builds only, tools/ab_build.py: the epilogue and the x stream alone - wrong keys by design)
#pragma unroll
    for (int t = 0; t < KT; ++t) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][t % NCT] += xr[rt][t][0] + xr[rt][t][1];
      load_x(next, t);
    }
#else
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      Bf16Pairs hi[RT], mid[RT];
      if (32 * (t + 1) > dim) {                                        // (uniform) a k-tile that reaches past the row's end
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bool gone = 32 * t + 8 * g + 4 * c >= dim;
#pragma unroll
            for (int e = 0; e < 4; ++e) xr[rt][t][c][e] = gone ? 0.f : xr[rt][t][c][e];
          }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {                               // the split of sig16_kernel: hi = bf16(x), mid = bf16(x - hi)
          const float v0 = xr[rt][t][pr >> 1][2 * (pr & 1)], v1 = xr[rt][t][pr >> 1][2 * (pr & 1) + 1];
          const bf16x2 hp = bf16x2{(__bf16)v0, (__bf16)v1};
          const float r0 = v0 - (float)hp[0], r1 = v1 - (float)hp[1];
          const bf16x2 mp = bf16x2{(__bf16)r0, (__bf16)r1};
          hi[rt].p[pr] = hp;
          mid[rt].p[pr] = mp;
          ss[rt] = __builtin_amdgcn_fdot2_f32_bf16(hp, hp, ss[rt], false);
          sm[rt] = __builtin_amdgcn_fdot2_f32_bf16(mp, mp, sm[rt], false);
          asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax[rt]) : "v"(v0), "v"(v1));
        }
      load_x(next, t);                                                 // this k-tile's registers are free: the next tile's elements
      const f32x4* frag = reinterpret_cast<const f32x4*>(lds) + (size_t)t * NCT * 128 + lane;
#pragma unroll
      for (int cp = 0; cp < NCT / 2; ++cp) {                           // two column tiles at a time: four accumulators take turns
        f32x4 ph[2], pm[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          ph[j] = frag[(2 * cp + j) * 128];
          pm[j] = frag[(2 * cp + j) * 128 + 64];
        }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
              const bf16x8 av = __builtin_bit_cast(bf16x8, term == 2 ? mid[rt] : hi[rt]);
              const bf16x8 bv = __builtin_bit_cast(bf16x8, term == 1 ? pm[j] : ph[j]);
              acc[rt][2 * cp + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[rt][2 * cp + j], 0, 0, 0);   // D = P X^T: (column, row)
            }
      }
    }

#endif
#ifndef LSHRS_AB_RES_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef LSHRS_AB_RES_PROBE
    asm volatile("" ::: "memory");
    const unsigned long long pr_t1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef LSHRS_AB_RES_NO_EPILOGUE   // (A/B builds only: the main loop alone - one word per lane keeps the accumulators alive)
    {
      float keep = 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) keep += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
      keep += ss[0] + sm[0] + amax[0];
      if (keep == 123.456f) args.keys[lane] = 1;
    }
    continue;
#endif
    // ---- row statistics -> the two factors of the stage-1 window (as sig16_kernel).  The accumulators are TRANSPOSED
    // (D = P X^T: lane (r16, g) holds columns 4 g .. + 3 of every column tile for row r16), so a lane needs the factors of
    // ONE row - its own - and has them in registers after the reduction: no trip through LDS.
    float wnd[RT], wnb[RT], tsmax[RT];
    bool zrow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float s2 = ss[rt] + __shfl_xor(ss[rt], 16);
      s2 += __shfl_xor(s2, 32);
      float m2 = sm[rt] + __shfl_xor(sm[rt], 16);
      m2 += __shfl_xor(m2, 32);
      float am = __builtin_fmaxf(amax[rt], __shfl_xor(amax[rt], 16));
      am = __builtin_fmaxf(am, __shfl_xor(am, 32));
      const int64_t myrow = row0 + 16 * rt + r16;
      // (v_sqrt_f32 as it is - 1 ulp - instead of the library's corrected root: the 0.1 % below covers far more)
      float window = __builtin_amdgcn_sqrtf(s2) * args.tau * 1.001f;
      if (am != 0.f && !(am >= 0x1p-32f && am <= 0x1p32f)) window = __builtin_inff();
      float wb_ = __builtin_amdgcn_sqrtf(m2) * args.tau_b * 1.001f;
      wb_ = wb_ < __builtin_inff() ? wb_ : 0.f;
      if (g == 0 && args.row_flags != nullptr && myrow < args.n) {
        const bool has_nan = s2 != s2;
        const bool zero = (am <= 1e-8f) && !has_nan;
        args.row_flags[myrow] = (uint8_t)((zero ? 1 : 0) | (has_nan ? 2 : 0));
      }
      wnd[rt] = window;
      wnb[rt] = wb_;
      float ts = window < __builtin_inff() ? window * amax_cb + wb_ * bmax_cb : __builtin_inff();
      tsmax[rt] = ts > 0.f ? ts : -1.f;                  // a zero row: nothing to re-evaluate
      zrow[rt] = am == 0.f;                              // every projection of the row is +0: all bits 0 (a NaN row is flagged wholesale)
    }

    // the audit sample of this tile (SigArgs::audit_list): one of its 16 RT x 16 NCT projections
    int au_slot = -1, au_rw = -1, au_lane = 0, au_q = 0;
    if (args.audit_list != nullptr && (int)(tile % args.audit_div) == args.audit_phase) {
      const unsigned h = audit_hash((unsigned)tile, args.audit_seed);
      au_slot = (int)(tile / args.audit_div);
      au_rw = (int)((h & 1u) % (unsigned)RT) * 8 + (int)(((h >> 1) & 7u) % (unsigned)NW);
      au_q = (int)((h >> 4) & 7u);
      au_lane = (int)((h >> 7) & 63u);
    }

    // ---- sign bits, window test, list ------------------------------------------------------------------------------
    // A lane's 4 NCT values of a row are 4 NCT consecutive bits of the row's sign string (res_colmap: the image is packed in
    // that order): ONE v_alignbit per value shifts the accumulator's sign into the lane's word - the string holds y < 0, the
    // key wants y > 0: the word is inverted on its way out (exact zeros are flagged, or the whole row is zero).
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        uint32_t word[kResGroups] = {};
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          float m = __builtin_inff();
#pragma unroll
          for (int reg = 0; reg < 4; reg += 2) {
            const float y0 = acc[rt][2 * w][reg], y1 = acc[rt][2 * w][reg + 1];
            word[w >> 2] = __builtin_amdgcn_alignbit(word[w >> 2], __float_as_uint(y0), 31u);
            word[w >> 2] = __builtin_amdgcn_alignbit(word[w >> 2], __float_as_uint(y1), 31u);
            asm("v_min3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(y0), "v"(y1));        // (NaN dropped)
          }
#pragma unroll
          for (int reg = 0; reg < 4; reg += 2) {
            const float y0 = acc[rt][2 * w + 1][reg], y1 = acc[rt][2 * w + 1][reg + 1];
            word[w >> 2] = __builtin_amdgcn_alignbit(word[w >> 2], __float_as_uint(y0), 31u);
            word[w >> 2] = __builtin_amdgcn_alignbit(word[w >> 2], __float_as_uint(y1), 31u);
            asm("v_min3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(y0), "v"(y1));
          }
          const bool aud = au_rw == 8 * rt + w;
          if (__builtin_amdgcn_ballot_w64(!(m > tsmax[rt])) != 0 || aud) {   // wave-uniform: the exact per-element test
            unsigned hits = 0u;
            float ys[8], thrs[8];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int ct = 2 * w + half;
              const f32x4 pa = *reinterpret_cast<const f32x4*>(coef_lds + 16 * ct + 4 * g);
              const f32x4 pb = *reinterpret_cast<const f32x4*>(coef_lds + 256 + 16 * ct + 4 * g);
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                float thr = wnd[rt] * pa[reg] + wnb[rt] * pb[reg];
                thr = thr > 0.f ? thr : -1.f;                             // zero row / zero-padded column: y is exactly 0
                ys[4 * half + reg] = acc[rt][ct][reg];
                thrs[4 * half + reg] = thr;
                hits |= (!(__builtin_fabsf(ys[4 * half + reg]) > thr) ? 1u : 0u) << (4 * half + reg);
              }
            }
            const int64_t grow = row0 + 16 * rt + r16;
            if (aud && lane == au_lane) {
              float yq = ys[0], tq = thrs[0];
#pragma unroll
              for (int q = 1; q < 8; ++q) {
                yq = au_q == q ? ys[q] : yq;
                tq = au_q == q ? thrs[q] : tq;
              }
              const int colid = padcol_lds[16 * (2 * w + (au_q >> 2)) + 4 * g + (au_q & 3)];
              const bool keep = ((hits >> au_q) & 1u) == 0u && grow < args.n && colid >= 0 && tq < __builtin_inff();
              args.audit_list[au_slot] = keep ? ((grow << 21) | (int64_t)colid) : (int64_t)-1;
              args.audit_vals[2 * au_slot] = yq;
              args.audit_vals[2 * au_slot + 1] = tq;
            }
            if (hits != 0u) {
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const int reg = q & 3, ct = 2 * w + (q >> 2);
                const int colid = padcol_lds[16 * ct + 4 * g + reg];
                if (((hits >> q) & 1u) != 0u && grow < args.n && colid >= 0) {
                  const int64_t entry = (grow << 21) | (int64_t)colid;
                  const float ykeep = wnd[rt] < __builtin_inff() ? ys[q] : __builtin_nanf("");
                  const int pos = atomicAdd(l_count, 1);                  // LDS atomic on the wave's own counter
                  if (pos < kResListCap) {
                    l_list[pos] = entry;
                    l_y[pos] = ykeep;
                  } else {                                                // the wave's stage is full (rows flagged wholesale)
                    const int slot = atomicAdd(args.tie_count, 1);
                    if (slot < args.tie_cap) {
                      args.tie_list[slot] = entry;
                      if (args.flag_y != nullptr) args.flag_y[slot] = ykeep;
                    }
                  }
                }
              }
            }
          }
        }
        // the lane's bits of row 16 rt + r16: group k of eight column tiles = words 4 k + g of the row's string (32 bits a
        // lane), a last group of four = half-words
        const int rl = 16 * rt + r16;
#pragma unroll
        for (int k = 0; k < kResGroups; ++k) {
          const uint32_t v = zrow[rt] ? 0u : ~word[k];
          if (NCT - 8 * k >= 8) cw_lds[rl * 9 + 4 * k + g] = v;
          else reinterpret_cast<uint16_t*>(cw_lds)[(rl * 9 + 4 * k) * 2 + g] = (uint16_t)v;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- key bytes: byte o of a row = bits [src, src + 8) of its sign string, masked to the band's live rows ----------
    constexpr int kLPR = 64 / kRows;                      // lanes per row of the tile
    if (args.vec_store) {                                 // whole 32-bit words of 4-byte aligned key rows
      const int nw32 = nby >> 2, rl = lane / kLPR;
      for (int o4 = lane % kLPR; o4 < nw32; o4 += kLPR) {
        uint32_t out = 0u;
        const u32x4 rec = *reinterpret_cast<const u32x4*>(tab_lds + 4 * o4);       // the four bytes' table entries
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int src = (int)(rec[b] & 0xFFFFu), w = src >> 5;
          const uint32_t lo = cw_lds[rl * 9 + w], hi_ = cw_lds[rl * 9 + w + 1];    // (a ninth word per row: never live, always readable)
          const uint32_t v = __builtin_amdgcn_alignbit(hi_, lo, (uint32_t)(src & 31)) & (rec[b] >> 16);
          out |= v << (8 * b);
        }
#ifdef LSHRS_AB_RES_NO_KEYSTORE     // (A/B builds only: what the key stores cost the waves' load waits - one word per launch keeps the work alive)
        if (out == 0x12345678u && row0 + rl < args.n) *reinterpret_cast<uint32_t*>(args.keys + (row0 + rl) * (int64_t)nby + 4 * o4) = out;
#else
        if (row0 + rl < args.n) *reinterpret_cast<uint32_t*>(args.keys + (row0 + rl) * (int64_t)nby + 4 * o4) = out;
#endif
      }
    } else {
      const int rl = lane / kLPR;
      for (int o = lane % kLPR; o < nby; o += kLPR) {
        const uint32_t rec = (uint32_t)tab_lds[o];
        const int src = (int)(rec & 0xFFFFu), w = src >> 5;
        const uint32_t lo = cw_lds[rl * 9 + w], hi_ = cw_lds[rl * 9 + w + 1];
        const uint32_t v = __builtin_amdgcn_alignbit(hi_, lo, (uint32_t)(src & 31)) & (rec >> 16);
        if (row0 + rl < args.n) args.keys[(row0 + rl) * (int64_t)nby + o] = (uint8_t)v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- the wave's staged list: out when it is half full or the wave is done --------------------------------------
    {
      const int staged_raw = l_count[0];
      const int staged = staged_raw < kResListCap ? staged_raw : kResListCap;
      if (staged >= kResListCap / 2 || (tile + stride >= tiles && staged > 0)) {       // (wave-uniform)
        int base = 0;
        if (lane == 0) base = atomicAdd(args.tie_count, staged);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int e = lane; e < staged; e += 64) {
          const int slot = base + e;
          if (slot < args.tie_cap) {
            args.tie_list[slot] = l_list[e];
            if (args.flag_y != nullptr) args.flag_y[slot] = l_y[e];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) l_count[0] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
#ifdef LSHRS_AB_RES_PROBE
    {
      asm volatile("" ::: "memory");
      const unsigned long long pr_t2 = __builtin_amdgcn_s_memtime();
      pr_main += pr_t1 - pr_t0;
      pr_epi += pr_t2 - pr_t1;
      pr_tiles += 1;
    }
#endif
  }
#ifdef LSHRS_AB_RES_PROBE
  if (args.clock_probe != nullptr && lane == 0) {
    unsigned long long* q = args.clock_probe + 6 * ((size_t)blockIdx.x * kResWaves + wave);
    q[0] = pr_main; q[1] = pr_epi; q[2] = pr_tiles;
    q[3] = __builtin_amdgcn_s_memtime() - pr_t00; q[4] = __builtin_amdgcn_s_memrealtime() - pr_r00; q[5] = 1;
  }
#endif
}

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

template <bool ALIGNED>
__global__ __launch_bounds__(kCosThreads) void cosine_kernel(const float* __restrict__ corpus, int64_t m, int64_t ldc,
                                                             int dim, const float* __restrict__ queries,
                                                             const int64_t* __restrict__ cand_idx, int c, int slices,
                                                             float* __restrict__ scores, uint8_t* __restrict__ status,
                                                             uint8_t* __restrict__ qstatus) {
  extern __shared__ __attribute__((aligned(16))) float qlds[];  // dim floats (+ pad to 4) + kCosWaves partials
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int qi = blockIdx.x / slices;
  const int slice = blockIdx.x % slices;
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
        idx = cand_idx != nullptr ? cand_idx[(int64_t)qi * c + ci] : (int64_t)qi * c + ci;
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
        const int64_t o = (int64_t)qi * c + base + u;
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

// ==========================================================================================
// C ABI
// ==========================================================================================
namespace {
struct Opts {            // the caller's lshrs_sig_opts, or all-null
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long* clock_probe = nullptr;
  const lshrs_sig_sort* sort = nullptr;
};
inline Opts read_opts(const lshrs_sig_opts* o) {
  Opts r;
  if (o != nullptr && o->struct_bytes >= offsetof(lshrs_sig_opts, sort)) {       // (a caller built against the struct without `sort`)
    r.ev[0] = static_cast<hipEvent_t>(o->ev_stage1_start);
    r.ev[1] = static_cast<hipEvent_t>(o->ev_stage1_stop);
    r.ev[2] = static_cast<hipEvent_t>(o->ev_stage2_start);
    r.ev[3] = static_cast<hipEvent_t>(o->ev_stage2_stop);
    r.clock_probe = static_cast<unsigned long long*>(o->clock_probe);
    if (o->struct_bytes >= sizeof(lshrs_sig_opts) && o->sort != nullptr && o->sort->struct_bytes >= sizeof(lshrs_sig_sort) &&
        o->sort->list != nullptr && o->sort->y != nullptr && o->sort->hist != nullptr)
      r.sort = o->sort;
  }
  return r;
}
}  // namespace

extern "C" {

int lshrs_abi_version(void) { return LSHRS_ABI_VERSION; }

// Which measurement switches this build was compiled with (include/lshrs_hip.h, LSHRS_BUILD_*): the product build returns 0.
uint32_t lshrs_build_flags(void) {
  uint32_t f = 0;
#ifdef LSHRS_AB_FIX_SAME_P
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 8);
#endif
#ifdef LSHRS_AB_FIX_NO_X
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 9);
#endif
#ifdef LSHRS_AB_FIX_NO_P
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 10);
#endif
#ifdef LSHRS_AB_NO_XMID_NORM
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 11);
#endif
#ifdef LSHRS_AB_RES_L2ROWS
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 12);
#endif
#ifdef LSHRS_AB_RES_NO_MAIN
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 13);
#endif
#ifdef LSHRS_AB_RES_NO_EPILOGUE
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 14);
#endif
#ifdef LSHRS_AB_RES_NO_KEYSTORE
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 15);
#endif
#ifdef LSHRS_AB_NO_STATIC_PRIO
  f |= LSHRS_BUILD_TUNED | (1u << 16);
#endif
#ifdef LSHRS_AB_RES_COPY_PROLOGUE
  f |= LSHRS_BUILD_TUNED | (1u << 17);
#endif
#ifdef LSHRS_AB_RES_NO_PRIO
  f |= LSHRS_BUILD_TUNED | (1u << 18);
#endif
#ifdef LSHRS_AB_RES_PROBE
  f |= LSHRS_BUILD_TUNED | (1u << 19);
#endif
#ifdef LSHRS_T16_BUILTIN
  f |= LSHRS_BUILD_TUNED | (1u << 20);
#endif
#if LSHRS_X_AUX != 0 || LSHRS_FIX_SLAB != 6 || LSHRS_FIX_GRID != 1536 || LSHRS_RES_RT2_MAX != 16 || LSHRS_RES_WAVES_32 != 12 || \
    (defined(LSHRS_RES_GRID) && LSHRS_RES_GRID != 256)
  f |= LSHRS_BUILD_TUNED | (1u << 21);
#endif
  return f;
}

static bool sig_shape_ok(int32_t num_bands, int32_t rows, int32_t dim) {
  if (num_bands <= 0 || rows <= 0 || dim <= 0) return false;
  const int64_t padcols = (int64_t)num_bands * ((rows + 7) / 8) * 8;
  return padcols <= (1 << 21);
}

int64_t lshrs_sig_workspace_bytes(int32_t num_bands, int32_t rows_per_band, int32_t dim) {
  if (!sig_shape_ok(num_bands, rows_per_band, dim)) return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  return sig_workspace_floats(g, num_bands, rows_per_band, dim) * (int64_t)sizeof(float);
}

int32_t lshrs_sig_padded_columns(int32_t num_bands, int32_t rows_per_band) {
  if (num_bands <= 0 || rows_per_band <= 0) return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, 1);
  return g.cb * g.nt * 32;
}

int lshrs_sig_pack_projections(const float* P, int32_t num_bands, int32_t rows_per_band, int32_t dim, void* workspace,
                               void* stream) {
  if (P == nullptr || workspace == nullptr || !sig_shape_ok(num_bands, rows_per_band, dim)) return LSHRS_E_BADARG;
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* image = static_cast<float*>(workspace);
  float* norms = image + sig_image_floats(g);
  const int64_t chunks = sig_image_floats(g) / 4;
  hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                     rows_per_band, dim, g.bb, g.nt, g.ktiles, chunks, reinterpret_cast<f32x4*>(image));
  const int cols = g.cb * g.nt * 32;
  hipLaunchKernelGGL(pack_norm_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), 0, s, P, num_bands, rows_per_band,
                     dim, g.bb, cols, norms);
  hipLaunchKernelGGL(pack_normmax_kernel, dim3((unsigned)((g.cb + 63) / 64)), dim3(64), 0, s, norms, g.nt * 32, g.cb,
                     norms + sig_norm_floats(g));
  if (sig_has_fine(g)) {
    const SigGeom f = sig_fine_geom(g);
    float* fimage = image + sig_main_floats(g);
    const int64_t fchunks = sig_image_floats(f) / 4;
    hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)((fchunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, f.bb, f.nt, f.ktiles, fchunks, reinterpret_cast<f32x4*>(fimage));
    hipLaunchKernelGGL(pack_normmax_kernel, dim3((unsigned)((f.cb + 63) / 64)), dim3(64), 0, s, norms, 32, f.cb,
                       fimage + sig_image_floats(f));
  }
  if (sig_has_narrow_split(g)) {
    float* nimage = image + sig_narrow_offset_floats(g);
    const int64_t nchunks = sig_narrow_image_floats(g) / 4;
    hipLaunchKernelGGL(pack_image_bf16_t16_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, g.bb, g.ktiles, nchunks, reinterpret_cast<u16x8*>(nimage));
    float* nnorms = nimage + sig_narrow_image_floats(g);
    hipLaunchKernelGGL(pack_norm_kernel, dim3(4), dim3(64), 0, s, P, num_bands, rows_per_band, dim, g.bb, 256, nnorms);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), dim3(64), 0, s, nnorms, 256, 1, nnorms + 256);
  }
  {
    const int ldp = g.ktiles * kKTile;
    const int64_t total = (int64_t)cols * ldp;
    hipLaunchKernelGGL(pack_rowmajor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, g.bb, cols, ldp, image + sig_rowmajor_offset_floats(g));
  }
  if (sig_has_split(g)) {
    const int64_t schunks = sig_image_floats(g) / 4;  // 16-byte chunks: same count as the f32 image
    float* timage = image + sig_t16_offset_floats(g);   // hi / mid bf16 parts in 16x16x32 fragment order
    hipLaunchKernelGGL(pack_image_bf16_t16_kernel, dim3((unsigned)((schunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, g.bb, g.ktiles, schunks, reinterpret_cast<u16x8*>(timage));
  }
  {
    // the window block until lshrs_sig_set_window fills it: coefficients so large that a pass asking for the proven
    // window sends EVERY projection to the exact decision (slow and right, never fast and wrong)
    const int64_t wf = sig_window_floats(g);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((wf + 255) / 256)), dim3(256), 0, s, image + sig_window_offset_floats(g),
                       wf, 1e30f);
  }
  const SigCompact cp = sig_compact(g, num_bands, rows_per_band);
  if (cp.on) {                                      // stage 1's own image, tables and copies (sig_compact)
    const SigCompactWs cw = sig_compact_ws(image, g, cp);
    const int cc = cp.ncb * 256;
    const int64_t cchunks = (int64_t)cp.ncb * g.ktiles * 8192 / 4;
    hipLaunchKernelGGL(compact_tables_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, num_bands, rows_per_band,
                       g.bb, cp.bpb, cp.ncb, cw.padcol, cw.bytetab);
    hipLaunchKernelGGL(pack_image_bf16_t16_kernel, dim3((unsigned)((cchunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, g.bb, g.ktiles, cchunks, reinterpret_cast<u16x8*>(cw.image), cp.bpb);
    hipLaunchKernelGGL(compact_gather_kernel, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, norms, cw.padcol, cc, cw.norms);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3((unsigned)((cp.ncb + 63) / 64)), dim3(64), 0, s, cw.norms, 256, cp.ncb, cw.norm_max);
    const int64_t wf = 2 * (int64_t)cc + 2 * sig_pad4(cp.ncb);          // wa_c .. wbmax_c are contiguous
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((wf + 255) / 256)), dim3(256), 0, s, cw.wa, wf, 1e30f);
  }
  const SigResident rs = sig_resident(num_bands, rows_per_band, dim);
  if (rs.on) {                                      // sig16r_kernel's image, tables and copies (sig_resident)
    const SigCompactWs rw = sig_resident_ws(image, g, num_bands, rows_per_band, rs);
    const int64_t rchunks = (int64_t)rs.kt * 8192 / 4;
    hipLaunchKernelGGL(compact_tables_kernel, dim3(1), dim3(256), 0, s, num_bands, rows_per_band, g.bb, num_bands, 1, rw.padcol,
                       rw.bytetab, rs.nct);
    hipLaunchKernelGGL(pack_image_bf16_t16_kernel, dim3((unsigned)((rchunks + 255) / 256)), dim3(256), 0, s, P, num_bands,
                       rows_per_band, dim, g.bb, rs.kt, rchunks, reinterpret_cast<u16x8*>(rw.image), num_bands, rs.nct);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(1), dim3(256), 0, s, norms, rw.padcol, 256, rw.norms);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), dim3(64), 0, s, rw.norms, 256, 1, rw.norm_max);
    const int64_t wf = 2 * 256 + 2 * 4;                                  // wa .. wbmax are contiguous
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((wf + 255) / 256)), dim3(256), 0, s, rw.wa, wf, 1e30f);
  }
  return -(int)hipGetLastError();
}

int lshrs_sig_set_window(void* workspace, int32_t num_bands, int32_t rows_per_band, int32_t dim, const float* coef_a,
                         const float* coef_b, const float* coef_tie, void* stream) {
  if (workspace == nullptr || coef_a == nullptr || coef_b == nullptr || coef_tie == nullptr ||
      !sig_shape_ok(num_bands, rows_per_band, dim))
    return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* base = static_cast<float*>(workspace);
  const SigWindow w = sig_window(base, g);
  const int wc = (int)sig_window_cols(g);
  hipLaunchKernelGGL(window_scatter_kernel, dim3((unsigned)((wc + 63) / 64)), dim3(64), 0, s, coef_a, coef_b, coef_tie,
                     num_bands, rows_per_band, g.bb, wc, const_cast<float*>(w.wa), const_cast<float*>(w.wb),
                     const_cast<float*>(w.wt));
  // maxima per column block of the main geometry (a narrow hasher has one block; its zero-padded tail adds nothing)
  const int per_block = g.nt * 32;
  const dim3 mg((unsigned)((g.cb + 63) / 64)), mb(64);
  hipLaunchKernelGGL(pack_normmax_kernel, mg, mb, 0, s, w.wa, per_block, g.cb, const_cast<float*>(w.wamax));
  hipLaunchKernelGGL(pack_normmax_kernel, mg, mb, 0, s, w.wb, per_block, g.cb, const_cast<float*>(w.wbmax));
  hipLaunchKernelGGL(pack_normmax_kernel, mg, mb, 0, s, w.wt, per_block, g.cb, const_cast<float*>(w.wtmax));
  const SigGeom f = sig_fine_geom(g);
  hipLaunchKernelGGL(pack_normmax_kernel, dim3((unsigned)((f.cb + 63) / 64)), dim3(64), 0, s, w.wt, 32, f.cb,
                     const_cast<float*>(w.wtmax_fine));
  if (sig_has_narrow_split(g)) {            // the narrow image is ONE block of 256 columns
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), dim3(64), 0, s, w.wa, 256, 1, const_cast<float*>(w.wamax) + kNarrowMaxSlot);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), dim3(64), 0, s, w.wb, 256, 1, const_cast<float*>(w.wbmax) + kNarrowMaxSlot);
  }
  const SigCompact cp = sig_compact(g, num_bands, rows_per_band);
  if (cp.on) {                              // stage 1's copies in the order of its compact column blocks
    const SigCompactWs cw = sig_compact_ws(base, g, cp);
    const int cc = cp.ncb * 256;
    const dim3 cg((unsigned)((cc + 255) / 256)), cbk(256), xg((unsigned)((cp.ncb + 63) / 64));
    hipLaunchKernelGGL(compact_gather_kernel, cg, cbk, 0, s, w.wa, cw.padcol, cc, cw.wa);
    hipLaunchKernelGGL(compact_gather_kernel, cg, cbk, 0, s, w.wb, cw.padcol, cc, cw.wb);
    hipLaunchKernelGGL(pack_normmax_kernel, xg, mb, 0, s, cw.wa, 256, cp.ncb, cw.wamax);
    hipLaunchKernelGGL(pack_normmax_kernel, xg, mb, 0, s, cw.wb, 256, cp.ncb, cw.wbmax);
  }
  const SigResident rs = sig_resident(num_bands, rows_per_band, dim);
  if (rs.on) {
    const SigCompactWs rw = sig_resident_ws(base, g, num_bands, rows_per_band, rs);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(1), dim3(256), 0, s, w.wa, rw.padcol, 256, rw.wa);
    hipLaunchKernelGGL(compact_gather_kernel, dim3(1), dim3(256), 0, s, w.wb, rw.padcol, 256, rw.wb);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), mb, 0, s, rw.wa, 256, 1, rw.wamax);
    hipLaunchKernelGGL(pack_normmax_kernel, dim3(1), mb, 0, s, rw.wb, 256, 1, rw.wbmax);
  }
  return -(int)hipGetLastError();
}

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

// A chunk of lshrs_sig_hash_batch_split_replay_chunked_f32 whose stage 2 runs BESIDE the next chunk's stage 1: stage 2 and the
// export are enqueued on `side` behind `ev_fork` (recorded on the pass's own stream behind stage 1), `ev_join` is recorded
// behind them; the caller makes its stream wait for ev_join before it returns.
struct SplitFork {
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
};

// blas_model 0: ties are reported in tie_list (the caller resolves them on the host); > 0: stage 2 resolves them itself
// by replaying that summation order of the host BLAS (sig_fix8_kernel<true>), tie_list is not used.
// counters (replay only): the LSHRS_SIG_DEVICE_COUNTERS block; flag_y: the stage-1 value of every list entry (may be NULL).
static int split_pass(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                      int32_t rows_per_band, int32_t dim, uint8_t* keys, int64_t* tie_list, int32_t tie_cap,
                      int32_t* tie_count, float tau, uint8_t* row_flags, int64_t* flag_list, float* flag_y,
                      int32_t flag_cap, int32_t* flag_count, float tau1, int blas_model, int32_t* counters,
                      int32_t* host_counts, const lshrs_sig_audit* audit, const lshrs_sig_opts* opts, void* stream,
                      const SplitFork* fork = nullptr) {
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
  const int64_t row_tiles = (n + 255) / 256;
  const int64_t wgs = (row_tiles + 7) / 8 * 8 * g.cb;
  if (n >= ((int64_t)1 << 42) || wgs > 0x7fffffffLL || g.cb > 65535) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Opts o = read_opts(opts);
  const float* base = static_cast<const float*>(workspace);
  // (a partial last k-tile - dim % 32 != 0 - only with the replay: its stage 2 is the one that reads the chunks past a row's end as zero)
  const bool aligned = (dim % 32 == 0 || (blas_model != 0 && dim % 4 == 0 && (dim >= 32 || rs.on))) && (ldx % 4 == 0) &&
                       ldx < (1 << 20) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
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
  // the audit sample (lshrs_sig_audit): one unit in `div` - a wave of sig16_kernel, a 32-row tile of sig16r_kernel
  int audit_n = 0;
  if (audit != nullptr && audit->struct_bytes >= sizeof(lshrs_sig_audit) && audit->list != nullptr && audit->vals != nullptr &&
      audit->slots > 0 && audit->target > 0 && blas_model != 0) {
    const int res_rows = 16 * res_rt(rs.nct, rs.kt);
    const int64_t units = rs.on ? (n + res_rows - 1) / res_rows : (row_tiles + 7) / 8 * 8 * a.ncb * 8;
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
    const dim3 grid((unsigned)(tiles < (int64_t)LSHRS_RES_GRID * rwaves ? (tiles + rwaves - 1) / rwaves : LSHRS_RES_GRID), 1, 1), block(64 * rwaves, 1, 1);
#define LSHRS_RES(NCT_, KT_) hipExtLaunchKernelGGL((sig16r_kernel<NCT_, KT_>), grid, block, 0, s, o.ev[0], o.ev[1], 0, a)
    if (rs.kt == 2) {
      if (rs.nct == 4) LSHRS_RES(4, 2); else if (rs.nct == 8) LSHRS_RES(8, 2); else if (rs.nct == 12) LSHRS_RES(12, 2); else LSHRS_RES(16, 2);
    } else if (rs.kt == 4) {
      if (rs.nct == 4) LSHRS_RES(4, 4); else if (rs.nct == 8) LSHRS_RES(8, 4); else if (rs.nct == 12) LSHRS_RES(12, 4); else LSHRS_RES(16, 4);
    } else {
      if (rs.nct == 4) LSHRS_RES(4, 8); else LSHRS_RES(8, 8);
    }
#undef LSHRS_RES
  } else {
    const dim3 grid((unsigned)((row_tiles + 7) / 8 * 8 * a.ncb), 1, 1);
    const bool partial = dim % kKTile != 0;
    if (cp.on) {
      if (partial) hipExtLaunchKernelGGL((sig16_kernel<true, true>), grid, dim3(512, 1, 1), 0, s, o.ev[0], o.ev[1], 0, a);
      else hipExtLaunchKernelGGL((sig16_kernel<true, false>), grid, dim3(512, 1, 1), 0, s, o.ev[0], o.ev[1], 0, a);
    } else {
      if (partial) hipExtLaunchKernelGGL((sig16_kernel<false, true>), grid, dim3(512, 1, 1), 0, s, o.ev[0], o.ev[1], 0, a);
      else hipExtLaunchKernelGGL((sig16_kernel<false, false>), grid, dim3(512, 1, 1), 0, s, o.ev[0], o.ev[1], 0, a);
    }
  }
  // stage 2: the flagged projections, one by one (a forked chunk: on the side stream, behind stage 1's event)
  if (fork != nullptr) {
    hipError_t e = hipEventRecord(fork->ev_fork, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(fork->side, fork->ev_fork, 0);
    if (e != hipSuccess) return -(int)e;
    s = fork->side;
  }
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
  const int64_t groups = ((int64_t)flag_cap + kFixG - 1) / kFixG;
  const bool short_rows = g.ktiles <= kFixSlabShort && blas_model != 0;
  const int grid_cap = short_rows ? kFixGridShort : kFixGridG;
  const dim3 grid((unsigned)(groups < grid_cap ? groups : grid_cap)), block(64);
  if (blas_model != 0) {
    f.tie_list = nullptr;
    f.flag_y = flag_y;
    f.partials = counters + LSHRS_SIG_COUNTERS;
    f.count_ties = 1;
    f.audit_list = audit_n > 0 ? a.audit_list : nullptr;
    f.audit_vals = a.audit_vals;
    f.audit_n = audit_n;
    int nparts = (int)grid.x;
    const bool sorted = o.sort != nullptr && !short_rows && f.padcols <= kSortMaxCols &&
                        (int64_t)o.sort->cap >= (int64_t)flag_cap + (int64_t)kFixG * f.padcols;
    if (short_rows) {
      if (blas_general(rows_per_band, g.ktiles, dim))
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, true, kFixSlabShort>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
      else
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, false, kFixSlabShort>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    } else if (sorted) {
      // the list by column first (three launches), then stage 2 with ONE hyperplane per group of eight; the audit sample -
      // unsorted, a few thousand entries - through the plain instantiation behind it, its statistics in the slots behind
      int* hist = o.sort->hist;
      int* sorted_count = hist + (size_t)kSortWgs * f.padcols;
      hipExtLaunchKernelGGL(fix_sort_hist_kernel, dim3(kSortWgs), dim3(kSortThreads), 0, s, o.ev[2], nullptr, 0, flag_list, flag_count,
                            flag_cap, f.padcols, hist);
      hipLaunchKernelGGL(fix_sort_scan_kernel, dim3(1), dim3(kSortMaxCols), 0, s, hist, f.padcols, o.sort->list, sorted_count);
      hipLaunchKernelGGL(fix_sort_scatter_kernel, dim3(kSortWgs), dim3(kSortThreads), 0, s, flag_list, flag_y, flag_count, flag_cap,
                         f.padcols, hist, o.sort->list, o.sort->y);
      FixArgs fs = f;
      fs.sorted_list = o.sort->list;
      fs.sorted_y = flag_y != nullptr ? o.sort->y : nullptr;
      fs.sorted_count = sorted_count;
      fs.audit_list = nullptr;
      fs.audit_n = 0;
      const bool has_audit = f.audit_list != nullptr && f.audit_n > 0;
      const int64_t sgroups = ((int64_t)flag_cap + kFixG - 1) / kFixG + f.padcols;
      constexpr int kSortedGrid = (LSHRS_SIG_DEVICE_COUNTERS - LSHRS_SIG_COUNTERS) / kFixParts - 512 < kFixGridG
                                      ? ((LSHRS_SIG_DEVICE_COUNTERS - LSHRS_SIG_COUNTERS) / kFixParts - 512) / 8 * 8 : kFixGridG;
      const dim3 sgrid((unsigned)(sgroups < kSortedGrid ? sgroups : kSortedGrid));     // (512 statistics slots stay for the audit launch)
      hipEvent_t stop = has_audit ? nullptr : o.ev[3];
      if (blas_general(rows_per_band, g.ktiles, dim))
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, true, kFixSlabG, true>), sgrid, block, 0, s, nullptr, stop, 0, fs);
      else
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, false, kFixSlabG, true>), sgrid, block, 0, s, nullptr, stop, 0, fs);
      nparts = (int)sgrid.x;
      if (has_audit) {
        FixArgs fa = f;
        fa.flag_cap = 0;                    // (no list entries: only the audit groups)
        fa.count_ties = 0;
        fa.partials = f.partials + (size_t)kFixParts * sgrid.x;
        const int agroups = (f.audit_n + kFixG - 1) / kFixG;
        const dim3 agrid((unsigned)(agroups < 512 ? agroups : 512));
        if (blas_general(rows_per_band, g.ktiles, dim))
          hipExtLaunchKernelGGL((sig_fix8_kernel<true, true>), agrid, block, 0, s, nullptr, o.ev[3], 0, fa);
        else
          hipExtLaunchKernelGGL((sig_fix8_kernel<true, false>), agrid, block, 0, s, nullptr, o.ev[3], 0, fa);
        nparts += (int)agrid.x;
      }
    } else if (blas_general(rows_per_band, g.ktiles, dim))
      hipExtLaunchKernelGGL((sig_fix8_kernel<true, true>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    else
      hipExtLaunchKernelGGL((sig_fix8_kernel<true, false>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kExportThreads), 0, s, counters, host_counts, nparts);
  } else {
    hipExtLaunchKernelGGL((sig_fix8_kernel<false, false>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
  }
  if (fork != nullptr) {
    const hipError_t e = hipEventRecord(fork->ev_join, s);
    if (e != hipSuccess) return -(int)e;
  }
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

static int lshrs_sig_hash_batch_split_replay_f32_impl(const float* X, int64_t n, int64_t ldx, const void* workspace,
                                                      int32_t num_bands, int32_t rows_per_band, int32_t dim, uint8_t* keys,
                                                      int32_t* counters, float tau, uint8_t* row_flags, int64_t* flag_list,
                                                      float* flag_y, int32_t flag_cap, float tau1, int32_t blas_model,
                                                      int32_t* host_counts, const lshrs_sig_audit* audit,
                                                      const lshrs_sig_opts* opts, void* stream, const SplitFork* fork) {
  const bool resident = sig_resident(num_bands, rows_per_band, dim).on;
  if (blas_model != 1 || dim % 4 != 0 || (dim < 32 && !resident) || (dim % 8 != 0 && dim > 4096) || counters == nullptr)
    return LSHRS_E_BADARG;
  return split_pass(X, n, ldx, workspace, num_bands, rows_per_band, dim, keys, nullptr, 0, counters + 0, tau, row_flags,
                    flag_list, flag_y, flag_cap, counters + 1, tau1, blas_model, counters, host_counts, audit, opts, stream, fork);
}

int lshrs_sig_hash_batch_split_replay_f32(const float* X, int64_t n, int64_t ldx, const void* workspace,
                                          int32_t num_bands, int32_t rows_per_band, int32_t dim, uint8_t* keys,
                                          int32_t* counters, float tau, uint8_t* row_flags, int64_t* flag_list,
                                          float* flag_y, int32_t flag_cap, float tau1, int32_t blas_model,
                                          int32_t* host_counts, const lshrs_sig_audit* audit, const lshrs_sig_opts* opts,
                                          void* stream) {
  return lshrs_sig_hash_batch_split_replay_f32_impl(X, n, ldx, workspace, num_bands, rows_per_band, dim, keys, counters, tau,
                                                    row_flags, flag_list, flag_y, flag_cap, tau1, blas_model, host_counts, audit,
                                                    opts, stream, nullptr);
}

int lshrs_sig_hash_batch_split_replay_chunked_f32(const float* X, int64_t n, int64_t ldx, const void* workspace,
                                                  int32_t num_bands, int32_t rows_per_band, int32_t dim, uint8_t* keys,
                                                  int32_t* counters, float tau, uint8_t* row_flags, int64_t* flag_list,
                                                  float* flag_y, float tau1, int32_t blas_model, int32_t* host_counts,
                                                  const lshrs_sig_audit* audit, const lshrs_sig_opts* opts,
                                                  const lshrs_sig_chunk_plan* plan, void* stream) {
  if (plan == nullptr || plan->struct_bytes < sizeof(lshrs_sig_chunk_plan) || plan->nchunks < 1 ||
      plan->nchunks > LSHRS_SIG_MAX_CHUNKS || counters == nullptr || host_counts == nullptr)
    return LSHRS_E_BADARG;
  const int nc = plan->nchunks;
  int64_t total = 0;
  for (int c = 0; c < nc; ++c) {
    if (plan->rows[c] <= 0 || plan->flag_cap[c] <= 0) return LSHRS_E_BADARG;
    if (c + 1 < nc && (plan->side_stream[c] == nullptr || plan->ev_fork[c] == nullptr || plan->ev_join[c] == nullptr ||
                       plan->side_stream[c] == stream))
      return LSHRS_E_BADARG;
    total += plan->rows[c];
  }
  if (total != n) return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  const int64_t row_bytes = (int64_t)num_bands * g.bb;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int64_t lo = 0, list_off = 0;
  int32_t slot_off = 0;
  const bool audit_on = audit != nullptr && audit->struct_bytes >= sizeof(lshrs_sig_audit) && audit->list != nullptr &&
                        audit->vals != nullptr && audit->slots > 0 && audit->target > 0;
  for (int c = 0; c < nc; ++c) {
    const int64_t rows = plan->rows[c];
    // the audit sample and its slots: every chunk its share by rows (the last one what is left)
    lshrs_sig_audit au{};
    if (audit_on) {
      au = *audit;
      const int32_t slots = c + 1 < nc ? (int32_t)((int64_t)audit->slots * rows / n) : audit->slots - slot_off;
      int32_t target = (int32_t)((int64_t)audit->target * rows / n);
      if (target < 1) target = 1;
      au.list = audit->list + slot_off;
      au.vals = audit->vals + 2 * (int64_t)slot_off;
      au.slots = slots;
      au.target = target < slots ? target : slots;
      au.seed = audit->seed + 0x9E3779B9u * (uint32_t)c;
      slot_off += slots;
    }
    // the measurement hooks: one quadruple of events per chunk, handed over back to back behind the struct's own
    lshrs_sig_opts op{};
    const lshrs_sig_opts* opp = nullptr;
    if (opts != nullptr && opts->struct_bytes >= sizeof(lshrs_sig_opts)) {
      op = *opts;
      if (plan->ev_timing != nullptr) {
        op.ev_stage1_start = plan->ev_timing[4 * c + 0];
        op.ev_stage1_stop = plan->ev_timing[4 * c + 1];
        op.ev_stage2_start = plan->ev_timing[4 * c + 2];
        op.ev_stage2_stop = plan->ev_timing[4 * c + 3];
      }
      if (c != 0) op.clock_probe = nullptr;
      opp = &op;
    }
    SplitFork fk{};
    const bool forked = c + 1 < nc;
    if (forked) {
      fk.side = static_cast<hipStream_t>(plan->side_stream[c]);
      fk.ev_fork = static_cast<hipEvent_t>(plan->ev_fork[c]);
      fk.ev_join = static_cast<hipEvent_t>(plan->ev_join[c]);
    }
    int32_t* cnt = counters + (int64_t)c * LSHRS_SIG_DEVICE_COUNTERS;
    const int rc = lshrs_sig_hash_batch_split_replay_f32_impl(
        X + lo * ldx, rows, ldx, workspace, num_bands, rows_per_band, dim, keys + lo * row_bytes, cnt, tau,
        row_flags != nullptr ? row_flags + lo : nullptr, flag_list + list_off, flag_y != nullptr ? flag_y + list_off : nullptr,
        plan->flag_cap[c], tau1, blas_model, host_counts + (int64_t)c * LSHRS_SIG_COUNTERS, audit_on ? &au : nullptr, opp, stream,
        forked ? &fk : nullptr);
    if (rc != 0) {
      // (chunks already enqueued run to their end: the caller's stream must still see them finish before buffers go away)
      for (int d = 0; d < c && d + 1 < nc; ++d) (void)hipStreamWaitEvent(s, static_cast<hipEvent_t>(plan->ev_join[d]), 0);
      return rc;
    }
    lo += rows;
    list_off += plan->flag_cap[c];
  }
  for (int c = 0; c + 1 < nc; ++c) {
    const hipError_t e = hipStreamWaitEvent(s, static_cast<hipEvent_t>(plan->ev_join[c]), 0);
    if (e != hipSuccess) return -(int)e;
  }
  return 0;
}

int lshrs_sig_resolve_ties_replay_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                                       int32_t rows_per_band, int32_t dim, uint8_t* keys, const int64_t* tie_list,
                                       int32_t tie_cap, int32_t* counters, float tau, int64_t* flag_list,
                                       int32_t flag_cap, int32_t blas_model, int32_t* host_counts, void* stream) {
  int32_t* tie_count = counters;                          // [0] tie entries the f32 kernel wanted to write
  int32_t* flag_count = counters != nullptr ? counters + 1 : nullptr;   // [1] items expanded from them
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || keys == nullptr || tie_list == nullptr || tie_count == nullptr ||
      flag_list == nullptr || flag_count == nullptr || tie_cap <= 0 || flag_cap <= 0 || n < 0 || ldx < dim ||
      !sig_shape_ok(num_bands, rows_per_band, dim) || (blas_model != 1 && blas_model != 2))
    return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  const int row_bytes = num_bands * g.bb;
  // stage 2 stages 16-byte chunks of 16-byte aligned rows (key rows may have any width: split_pass's comment); anything
  // else - dim % 4 elements of scalar tail (blas_model 1 / 2: how the host compiles it), rows that are only 4-byte aligned -
  // goes through the plain-load form of the same replay (sig_fixany_kernel)
  const bool fast = dim % 4 == 0 && dim >= 8 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && rows_per_band >= 2;
  const int body = dim & ~3;
  // (a band of ONE row is sdot on the host: modelled for every length, both builds - the plain-load form follows it)
  const bool one_row = rows_per_band == 1;
  if (!one_row && ((body % 8 != 0 && body > 4096) || (!fast && dim < 9))) return LSHRS_E_TOOLARGE;
  if (n >= ((int64_t)1 << 42) || (fast && blas_model != 1)) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* base = static_cast<const float*>(workspace);
  {
    const int threads = 256;
    const int blocks = tie_cap < 256 * 64 ? (tie_cap + threads - 1) / threads : 64;
    hipLaunchKernelGGL(expand_ties_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, tie_list, tie_count, tie_cap,
                       row_bytes * 8, flag_list, flag_cap, flag_count);
  }
  FixArgs f{};
  f.X = X;
  f.ldx = ldx;
  f.dim = dim;
  f.ktiles = g.ktiles;
  f.prow = base + sig_rowmajor_offset_floats(g);
  f.norms = base + sig_image_floats(g);
  f.keys = keys;
  f.row_bytes = row_bytes;
  f.padcols = row_bytes * 8;
  f.flag_list = flag_list;
  f.flag_count = flag_count;
  f.flag_cap = flag_cap;
  f.row_base = 0;
  f.tie_list = nullptr;
  f.tie_cap = 0;
  f.tie_count = nullptr;          // (the caller has the number of tie entries already; stage 2 only decides them)
  f.tau = tau > 0.f ? tau : 1.0f;
  f.tie_coef = tau > 0.f ? f.norms : sig_window(base, g).wt;
  f.blas_model = blas_model;
  f.rows_per_band = rows_per_band;
  f.band_cols = 8 * g.bb;
  f.partials = counters + LSHRS_SIG_COUNTERS;
  {
    const int64_t groups = ((int64_t)flag_cap + kFixG - 1) / kFixG;
    const dim3 grid((unsigned)(groups < kFixGridG ? groups : kFixGridG)), block(64);
    f.tail_model = blas_model;
    if (!fast) hipLaunchKernelGGL(sig_fixany_kernel, grid, block, 0, s, f);
    else if (blas_general(rows_per_band, g.ktiles, dim)) hipLaunchKernelGGL((sig_fix8_kernel<true, true>), grid, block, 0, s, f);
    else hipLaunchKernelGGL((sig_fix8_kernel<true, false>), grid, block, 0, s, f);
    hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kExportThreads), 0, s, counters, host_counts, (int)grid.x);
  }
  return -(int)hipGetLastError();
}

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

int lshrs_stream_synchronize(void* stream) { return -(int)hipStreamSynchronize(static_cast<hipStream_t>(stream)); }

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
  const bool wide = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t n16 = wide ? nbytes / 16 : 0;
  const int tail = (int)(nbytes - 16 * n16 > 0x7fffffff ? 0 : nbytes - 16 * n16);
  if (!wide && nbytes > 0x7fffffff) return LSHRS_E_TOOLARGE;
  const int64_t want = (n16 + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
  hipLaunchKernelGGL(copy_to_host_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16,
                     static_cast<const uint8_t*>(src) + 16 * n16, static_cast<uint8_t*>(dst) + 16 * n16, tail);
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
    hipLaunchKernelGGL(cosine_kernel<true>, grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_idx, c, slices,
                       scores, status, qstatus);
  else
    hipLaunchKernelGGL(cosine_kernel<false>, grid, block, shmem, s, corpus, m, ldc, dim, queries, cand_idx, c, slices,
                       scores, status, qstatus);
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
