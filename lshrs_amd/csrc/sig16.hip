// sig16.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// K1: stage 1 of the split-precision signature pass - bf16 x 3 on v_mfma_f32_16x16x32_bf16, proven window, audit sample.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
// Grid: one dimension, blockIdx.x -> (row tile, column block) with the column blocks of one row tile eight workgroup
// ids apart: workgroups are dealt round-robin over the 8 XCDs, so the `cb` passes over the same 256 rows run on the
// SAME XCD at about the same time and the second one reads x from that XCD's L2 instead of HBM (config 5: 512 key
// columns = two column blocks).
#ifndef LSHRS_X_AUX
#define LSHRS_X_AUX 0              // cache policy of stage 1's x loads (A/B builds: 2 = nt)
#endif
// flagged projections a workgroup stages in LDS before its ONE global append: 8192 for the 256-row workgroup, 4096 for the 128-row one
// COMPACT: the column blocks hold the bands' key columns side by side (sig_compact) - list entries and keys leave through
// the tables; a template parameter so that the padded layout's kernel is instruction for instruction what it was.
// PARTIAL: vectors that are not whole 32-element k-tiles (300-d, 100-d; round 5: any length, 301-d, 767-d) - in the last k-tile
// the 16-byte chunks that reach past a row's end are fetched from the row's last four elements instead (never a byte past the
// row): the chunk that holds the row's last dim % 4 elements is shifted into place when it is read back, the ones behind it
// read as zero.  Rows may start at any 4-byte address (the LDS-DMA takes it: tools/probes/lds_dma_align_probe.hip).
// W: waves of a workgroup = 32 W rows.  W = 8: one workgroup per CU (the ring is 144 KiB).  W = 4 (round 6, vectors of up to
// ~512 elements): TWO workgroups of 128 rows per CU, each with its own fragment ring and a two-deep x ring (80 KiB: the x
// tile a k-tile's second stage reads into registers is free again two stages before the tile after next lands in its place) -
// independent barriers, so one workgroup's prologue, epilogue and dispatch run under the other one's k-tiles, which at ten
// to sixteen k-tiles is 20-28 % of a 256-row workgroup's time.  Twice the fragment traffic from L2, the same x traffic, the
// same instruction schedule per wave, the same accumulation.
template <bool COMPACT, bool PARTIAL = false, int W = 8>
__global__ __launch_bounds__(64 * W, 8 / W) void sig16_kernel(const SigArgs args) {
  constexpr int RT = 2;
  constexpr int kRows = 32 * W;                   // rows of a workgroup
  constexpr int kXDepth = W == 8 ? 3 : 2;         // x tiles in the ring
  constexpr int kS1ListCap = W == 8 ? 8192 : 4096;
  constexpr int kWaveRows = 16 * RT;
  constexpr int kPP = 16 / W;                     // fragment pieces a wave stages per stage
  constexpr int kXPS = RT;                        // x pieces a wave stages per stage (2 RT per k-tile)
  constexpr int kE = 6 * RT;                      // MFMAs per eighth: 2 column tiles x 3 terms x RT row tiles
  constexpr int kSlices = 12 * RT;                // split slices per k-tile: 4 RT pairs x 3 steps
  constexpr int kPHalf = 16 * kFragFloats;        // floats of one fragment stage (16 blocks of 1 KiB)
  constexpr int kXTile = kRows * kKTile;          // floats of one x tile of the workgroup
  constexpr int kXWave = kWaveRows * kKTile;
  constexpr int kRingFloats = 3 * kPHalf + kXDepth * kXTile;
  constexpr int kIssues = (kPP + kXPS) / 2;       // DMA instructions a wave issues under each of a stage's last two eighths
  constexpr int kIssueStep = kE / kIssues;
  static_assert((kPP + kXPS) % 2 == 0 && kE % kIssues == 0, "DMA issue slots");
  static_assert(3 * kS1ListCap <= kRingFloats, "the epilogue's list stage reuses the ring");
  // the epilogue's tables: behind the list stage in the ring; the rows' two windows and the list counters behind the ring where
  // one workgroup owns the CU, inside it (behind the sign words) where two share it: 80 KiB each, to the byte
  constexpr int kTabOff = 3 * kS1ListCap;
  constexpr int kWndOff = W == 8 ? kRingFloats : kTabOff + 768 + kRows * 8;
  static_assert(kWndOff + 2 * kRows + 4 <= (W == 8 ? kRingFloats + 512 + 4 : kRingFloats), "windows and counters");
  __shared__ __attribute__((aligned(16))) float lds[W == 8 ? kRingFloats + 512 + 4 : kRingFloats];
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
    const unsigned u = (unsigned)bid * (unsigned)W + (unsigned)wave;
    if ((int)(u % (unsigned)args.audit_div) == args.audit_phase) {
      const unsigned h = audit_hash(u, args.audit_seed);
      au_slot = (int)(u / (unsigned)args.audit_div);
      au_rw = (int)(h & 15u);
      au_q = (int)((h >> 4) & 7u);
      au_lane = (int)((h >> 7) & 63u);
    }
  }
  if ((int64_t)row_tile * kRows >= args.n) {           // (whole workgroup: the grid is padded to a multiple of 8 row tiles)
    if (au_slot >= 0 && lane == 0) args.audit_list[au_slot] = -1;
    return;
  }
  const int ktiles = args.ktiles;
  const int stages = 2 * ktiles, lasts = stages - 1;
  const char* img = reinterpret_cast<const char*>(args.image) + (size_t)cb * ktiles * 32768;
  const int64_t blk_row0 = (int64_t)row_tile * kRows;
  const int64_t row0 = blk_row0 + wave * kWaveRows;
  const char* xblk = reinterpret_cast<const char*>(args.X + blk_row0 * args.ldx);

  // DMA offsets.  x is staged in FULL 128-byte lines: piece j (0..3) of a wave = rows 8j..8j+7 of its 32; lane
  // l = (r = l>>3, q = l&7) fetches 16-byte chunk q ^ r ^ (j&1) of row 8j + r, so an 8-lane group covers one whole
  // line (in permuted order) and the read-back (each lane: its row's two chunks of the k-tile quarter it feeds) is
  // conflict-free for ds_read_b128's 16-lane groups.
  unsigned poff[kPP], xfo[2 * RT], xrd[RT][2];
#pragma unroll
  for (int q = 0; q < kPP; ++q) poff[q] = (unsigned)(((W * (W == 8 ? q : 0) + wave) * 64 + lane) * 16);
  // (W = 4: one offset in a register, piece d is W d KiB behind it on both sides - four would not fit beside the accumulators)
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
  const int last_valid_chunks = PARTIAL ? (args.dim - (ktiles - 1) * kKTile) / 4 : 8;   // WHOLE 16-byte chunks of a row in the last k-tile
  const int rem = PARTIAL ? (args.dim & 3) : 0;                                          // elements of the chunk behind them
  const int tail_off = PARTIAL ? (args.dim - 4 - (ktiles - 1) * kKTile) * 4 : 0;         // the row's last four elements, from the last k-tile's start (bytes)
  auto plan = [&](int s) {          // what stage s issues: fragments of stage s+2, x pieces 4(s&1).. of tile (s>>1)+2
    Dma f;
    const int ns = s + 2, c = ns < lasts ? ns : lasts;
    const int nt = (s >> 1) + 2, t = nt < ktiles ? nt : ktiles - 1;
    f.pg = img + (size_t)c * 16384;
    f.xg = xblk + (size_t)t * (kKTile * 4);
    f.pdst = lds + (ns % 3) * kPHalf + wave * kFragFloats;
    f.j0 = kXPS * (s & 1);
    f.xdst = lds + 3 * kPHalf + (nt % kXDepth) * kXTile + wave * kXWave + f.j0 * kFragFloats;
    return f;
  };
  auto issue = [&](const Dma& f, int d) {
    if constexpr (!PARTIAL) {
      if (d < kPP)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.pg + (W == 8 ? poff[d] : poff[0] + (unsigned)(W * d * 1024))),
                                         (LDS_AS void*)(f.pdst + W * d * kFragFloats), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.xg + (f.j0 ? xfo[kXPS + d - kPP] : xfo[d - kPP])),
                                         (LDS_AS void*)(f.xdst + (d - kPP) * kFragFloats), 16, 0, LSHRS_X_AUX);
    } else if (d < kPP) {
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.pg + (W == 8 ? poff[d] : poff[0] + (unsigned)(W * d * 1024))),
                                       (LDS_AS void*)(f.pdst + W * d * kFragFloats), 16, 0, 0);
    } else {
      // branch-free (a branch here changes where hipcc joins the accumulator tiles around the inline-asm MFMAs): in the last
      // k-tile a chunk that is not wholly inside the row is fetched from the row's last four elements instead; this lane's
      // chunk of the line, as in xfo
      const unsigned off = f.j0 ? xfo[kXPS + d - kPP] : xfo[d - kPP];
      const int lim = f.xg == xblk + (size_t)(ktiles - 1) * (kKTile * 4) ? last_valid_chunks : 8;
      const int chunk = (int)((lane & 7) ^ (lane >> 3) ^ ((f.j0 + d - kPP) & 1));
      const int o2 = chunk >= lim ? (int)off - 16 * chunk + tail_off : (int)off;      // (off < 2^30: 256 rows of < 2^20 elements)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(f.xg + o2),
                                       (LDS_AS void*)(f.xdst + (d - kPP) * kFragFloats), 16, 0, LSHRS_X_AUX);
    }
  };
  f32x4 xr[RT][2];                           // raw f32 x of one k-tile: [row tile][chunk]
  auto read_x = [&](int t) {
    const char* xt = reinterpret_cast<const char*>(lds + 3 * kPHalf + (t % kXDepth) * kXTile + wave * kXWave);
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
          // the chunk with the row's last dim % 4 elements holds x[dim - 4 .. dim - 1]: element e is v[e + 4 - rem] for e < rem
          const bool part = rem != 0 && 2 * g + c == lim;
          const bool gone = 2 * g + c >= lim && !part;
          const f32x4 v = xr[rt][c];
          const float w0 = rem == 1 ? v[3] : (rem == 2 ? v[2] : v[1]);
          const float w1 = rem == 1 ? 0.f : (rem == 2 ? v[3] : v[2]);
          const float w2 = rem == 3 ? v[3] : 0.f;
          xr[rt][c] = part ? f32x4{w0, w1, w2, 0.f} : v;
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
  auto mfma_one = [&](int ct0, int k, const f32x4 (&f)[2][2], const Bf16Pairs (&hi)[RT], const Bf16Pairs (&mid)[RT], const bool drain = false) {
    const int term = k / (2 * RT), j = (k / RT) % 2, rt = k % RT;
    const bf16x8 a = __builtin_bit_cast(bf16x8, term == 2 ? mid[rt] : hi[rt]);
    const bf16x8 b = __builtin_bit_cast(bf16x8, f[j][term == 1 ? 1 : 0]);
    // Inline asm pins the accumulator to AGPRs and to in-place accumulation: left to the builtin, hipcc renames
    // accumulator tiles between MFMAs (vDst != SrcC) and parks some in VGPRs, i.e. hundreds of v_accvgpr moves and
    // s_nops per loop body.  Dependent MFMAs are four instructions (64 cycles) apart, beyond the 4-pass hazard window.
#ifdef LSHRS_T16_BUILTIN
    acc[rt][ct0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[rt][ct0 + j], 0, 0, 0);
#else
    // drain (behind the last barrier, where the two ends of the tile loop join): the wait states are PART of the statement -
    // hipcc may move accumulator tiles between registers at the join, and as two statements it put those moves between
    // the MFMA and its wait states (round 6, odd k-tile counts: the live audit fired; tools/check_mfma_hazards.py looks for it)
    if (drain) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 4" : "+a"(acc[rt][ct0 + j]) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[rt][ct0 + j]) : "v"(a), "v"(b));
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
      if (k % kIssueStep == 0) issue(f, k / kIssueStep);
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
      if (k % kIssueStep == 0) issue(f, kIssues + k / kIssueStep);
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
  if (W == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
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
    for (int k = 0; k < kE; ++k) mfma_one(14, k, fb, hs1, ms1, true);
  } else {
#pragma unroll
    for (int k = 0; k < kE; ++k) mfma_one(14, k, fb, hs0, ms0, true);
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
  int* l_count = reinterpret_cast<int*>(lds + kWndOff + 2 * kRows);   // [0] staged + overflowed entries, [1] global base
  if (tid == 0) l_count[0] = 0;

  // ---- row statistics -> the two factors of the stage-1 window per row ------------------------------------------------
  // ||x_hi|| and ||x_mid|| are sums over the very bf16 values the matrix instructions consumed (f32 accumulation: + 0.1 %,
  // which also covers what separates ||x_hi|| + ||x_mid|| from ||x||).  A row whose largest |x| is outside
  // [2^-32, 2^32] leaves the range in which the squares and the split neither underflow nor overflow: all of its
  // projections are re-evaluated (NOT(|y| > +inf) holds for every y).  A true zero row gives y = 0 in both passes.
  float* wnd_lds = lds + kWndOff + wave * kWaveRows;
  float* wnb_lds = lds + kWndOff + kRows + wave * kWaveRows;
  // the window coefficients of this column block, staged once (behind the list stage, which owns the first 3 x kS1ListCap
  // floats of the ring): with the proven window the exact test below runs on a third of the 32-column words, and a
  // global load in front of each of its compares is latency two waves per SIMD cannot hide
  float* coef_lds = lds + kTabOff;
  static_assert(kTabOff + 512 <= kRingFloats, "coefficients behind the list stage");
#pragma unroll
  for (int i = tid; i < 512; i += 64 * W) coef_lds[i] = i < 256 ? args.wa[cb * 256 + i] : args.wb[cb * 256 + i - 256];
  // compact column blocks (sig_compact): the padded id of every column of this block, and room for the block's sign words
  int* padcol_lds = reinterpret_cast<int*>(lds + kTabOff + 512);
  uint32_t* cw_lds = reinterpret_cast<uint32_t*>(lds + kTabOff + 768);
  static_assert(kTabOff + 768 + kRows * 8 <= kRingFloats, "compact tables behind the coefficients");
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
#ifdef LSHRS_AB_CHEAP_SIGNS      // (A/B builds only: what the transposition through SGPRs costs - one v_alignbit per value instead; wrong keys by design)
          A[w & 1] = __builtin_amdgcn_alignbit(A[w & 1], __float_as_uint(y0), 31u);
          B[w & 1] = __builtin_amdgcn_alignbit(B[w & 1], __float_as_uint(y1), 31u);
#else
          deposit_positive(A[w & 1], y0, 4 * p0 + (w >> 1), 4 * (p0 + 1) + (w >> 1));
          deposit_positive(B[w & 1], y1, 4 * p0 + (w >> 1), 4 * (p0 + 1) + (w >> 1));
#endif
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
            if (args.col_cap > 0) {           // buckets: the sample rides in its column's segment, marked
              if (keep) {
                const int slot = atomicAdd(args.col_count + colid, 1);
                if (slot < args.col_cap) {
                  const size_t at = (size_t)colid * args.col_cap + slot;
                  args.tie_list[at] = ((grow << 21) | (int64_t)colid) | kAuditBit;
                  args.flag_y[at] = yq;
                  args.flag_thr[at] = tq;
                }
              }
            } else {
            args.audit_list[au_slot] = keep ? ((grow << 21) | (int64_t)colid) : (int64_t)-1;
            args.audit_vals[2 * au_slot] = yq;
            args.audit_vals[2 * au_slot + 1] = tq;
            }
          }
          if (hits != 0u) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int reg = q & 3, ct = 2 * w + (q >> 2);
              const int64_t grow = row0 + 16 * rt + 4 * ge + reg;
              const int colid = COMPACT ? padcol_lds[16 * ct + r16e] : cb * 256 + 16 * ct + r16e;
              // (a NaN / Inf row also "flags" the zero-padded columns behind the last key column: the plain list carries
              //  them to stage 2, which skips them; a bucket launch has no segment for them)
              if (((hits >> q) & 1u) != 0u && grow < args.n && (!COMPACT || colid >= 0) &&
                  (args.col_cap == 0 || colid < args.row_bytes * 8)) {
                const int64_t entry = (grow << 21) | (int64_t)colid;
                // the stage-1 value travels with the entry: stage 2 measures |y1 - y_BLAS| on every flagged projection
                // (rows flagged wholesale carry no usable y1: NaN, skipped by that statistic)
                const float ykeep = wnd[reg] < __builtin_inff() ? ys[q] : __builtin_nanf("");
                const int pos = atomicAdd(l_count, 1);                    // LDS atomic
                if (pos < kS1ListCap) {
                  l_list[pos] = entry;
                  l_y[pos] = ykeep;
                } else if (args.col_cap > 0) {                            // LDS stage full, buckets: straight to the column's segment
                  atomicAdd(args.tie_count, 1);
                  const int slot = atomicAdd(args.col_count + colid, 1);
                  if (slot < args.col_cap) {
                    args.tie_list[(size_t)colid * args.col_cap + slot] = entry;
                    args.flag_y[(size_t)colid * args.col_cap + slot] = ykeep;
                  }
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
    for (int idx = tid; idx < kRows * nby; idx += 64 * W) {
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
  if (staged > 0 && args.col_cap > 0) {                 // buckets: every staged entry to its column's segment (one atomic on
    if (tid == 0) atomicAdd(args.tie_count, staged);    // one of <= 1024 addresses each: ~30 per workgroup at 768-d, ~60 at 1536-d)
    for (int e = tid; e < staged; e += 64 * W) {
      const int64_t entry = l_list[e];
      const int col = (int)(entry & ((1 << 21) - 1));
      const int slot = atomicAdd(args.col_count + col, 1);
      if (slot < args.col_cap) {
        args.tie_list[(size_t)col * args.col_cap + slot] = entry;
        args.flag_y[(size_t)col * args.col_cap + slot] = l_y[e];
      }
    }
  } else if (staged > 0) {                              // (workgroup-uniform)
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

}  // namespace

uint32_t lshrs_flags_sig16(void) {
  uint32_t f = 0;
#ifdef LSHRS_AB_NO_XMID_NORM
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 11);
#endif
#ifdef LSHRS_AB_NO_STATIC_PRIO
  f |= LSHRS_BUILD_TUNED | (1u << 16);
#endif
#ifdef LSHRS_AB_CHEAP_SIGNS
  f |= LSHRS_BUILD_WRONG_KEYS | (1u << 22);
#endif
#ifdef LSHRS_T16_BUILTIN
  f |= LSHRS_BUILD_TUNED | (1u << 20);
#endif
#if LSHRS_X_AUX != 0
  f |= LSHRS_BUILD_TUNED | (1u << 21);
#endif
  return f;
}

int lshrs_launch_sig16(const SigArgs& a, unsigned grid, bool compact, bool partial, bool half, hipStream_t s, hipEvent_t start,
                       hipEvent_t stop) {
  const dim3 g(grid, 1, 1), b(half ? 256 : 512, 1, 1);
  if (half) {          // 128-row workgroups, two per CU (`grid` counts those)
    if (compact) {
      if (partial) hipExtLaunchKernelGGL((sig16_kernel<true, true, 4>), g, b, 0, s, start, stop, 0, a);
      else hipExtLaunchKernelGGL((sig16_kernel<true, false, 4>), g, b, 0, s, start, stop, 0, a);
    } else {
      if (partial) hipExtLaunchKernelGGL((sig16_kernel<false, true, 4>), g, b, 0, s, start, stop, 0, a);
      else hipExtLaunchKernelGGL((sig16_kernel<false, false, 4>), g, b, 0, s, start, stop, 0, a);
    }
  } else if (compact) {
    if (partial) hipExtLaunchKernelGGL((sig16_kernel<true, true>), g, b, 0, s, start, stop, 0, a);
    else hipExtLaunchKernelGGL((sig16_kernel<true, false>), g, b, 0, s, start, stop, 0, a);
  } else {
    if (partial) hipExtLaunchKernelGGL((sig16_kernel<false, true>), g, b, 0, s, start, stop, 0, a);
    else hipExtLaunchKernelGGL((sig16_kernel<false, false>), g, b, 0, s, start, stop, 0, a);
  }
  return -(int)hipGetLastError();
}
