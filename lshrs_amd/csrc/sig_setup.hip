// sig_setup.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// The hyperplane workspace: re-layout kernels (f32 fragments, bf16 hi / mid fragments, row-major copy, norms, windows,
// compact and resident images) and the entry points that size and build it.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {

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
}  // namespace

extern "C" {

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

}  // extern "C"
