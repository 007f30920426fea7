// sig16r.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// K1r: stage 1 for short vectors - the whole fragment image resident in LDS, rows straight into registers, persistent waves.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
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
  // a wait for everything in flight behind it): a chunk past the row's end is fetched from the row's last four elements
  // instead and zeroed when its k-tile is split.  Rows of any length at any 4-byte address (round 5: 16 x 16 x 102, offset
  // views): the chunk that holds the row's last dim % 4 elements is ALSO fetched from the last four elements - never a byte
  // past the row - and shifted into place when its k-tile is split; the loads are 16 bytes wide at 4-byte alignment.
  struct __attribute__((packed, aligned(4))) Chunk { f32x4 v; };
  f32x4 xr[RT][KT][2];
  int koff[KT][2];
  const int rem = dim & 3;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) koff[t][c] = 32 * t + 8 * g + 4 * c + 4 <= dim ? 32 * t + 8 * g + 4 * c : dim - 4;
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
      for (int c = 0; c < 2; ++c) xr[rt][t][c] = reinterpret_cast<const Chunk*>(xp + koff[t][c])->v;
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
            const int k0 = 32 * t + 8 * g + 4 * c;
            const bool gone = k0 >= dim;
            if (rem != 0) {                                            // (uniform) the chunk with the row's last 1 .. 3 elements
              const bool part = !gone && k0 + 4 > dim;                 // holds x[dim - 4 .. dim - 1]: element e is x[k0 + e]
              const f32x4 v = xr[rt][t][c];                            // = v[e + 4 - rem] for e < rem, nothing behind it
              const float w0 = rem == 1 ? v[3] : (rem == 2 ? v[2] : v[1]);
              const float w1 = rem == 1 ? 0.f : (rem == 2 ? v[3] : v[2]);
              const float w2 = rem == 3 ? v[3] : 0.f;
              xr[rt][t][c] = part ? f32x4{w0, w1, w2, 0.f} : v;
            }
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

}  // namespace

uint32_t lshrs_flags_sig16r(void) {
  uint32_t f = 0;
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
#ifdef LSHRS_AB_RES_COPY_PROLOGUE
  f |= LSHRS_BUILD_TUNED | (1u << 17);
#endif
#ifdef LSHRS_AB_RES_NO_PRIO
  f |= LSHRS_BUILD_TUNED | (1u << 18);
#endif
#ifdef LSHRS_AB_RES_PROBE
  f |= LSHRS_BUILD_TUNED | (1u << 19);
#endif
#if LSHRS_RES_RT2_MAX != 16 || LSHRS_RES_WAVES_32 != 12
  f |= LSHRS_BUILD_TUNED | (1u << 21);
#endif
  return f;
}

int lshrs_launch_sig16r(const SigArgs& a, int nct, int kt, unsigned grid_x, unsigned block_x, hipStream_t s, hipEvent_t start,
                        hipEvent_t stop) {
  const dim3 grid(grid_x, 1, 1), block(block_x, 1, 1);
#define LSHRS_RES(NCT_, KT_) hipExtLaunchKernelGGL((sig16r_kernel<NCT_, KT_>), grid, block, 0, s, start, stop, 0, a)
  if (kt == 2) {
    if (nct == 4) LSHRS_RES(4, 2); else if (nct == 8) LSHRS_RES(8, 2); else if (nct == 12) LSHRS_RES(12, 2); else LSHRS_RES(16, 2);
  } else if (kt == 4) {
    if (nct == 4) LSHRS_RES(4, 4); else if (nct == 8) LSHRS_RES(8, 4); else if (nct == 12) LSHRS_RES(12, 4); else LSHRS_RES(16, 4);
  } else {
    if (nct == 4) LSHRS_RES(4, 8); else LSHRS_RES(8, 8);
  }
#undef LSHRS_RES
  return -(int)hipGetLastError();
}
