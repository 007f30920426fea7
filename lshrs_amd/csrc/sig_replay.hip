// sig_replay.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// Stage 2 of the signature pass: the host BLAS's summation order replayed for every flagged (and audited) projection -
// LDS-DMA form, column-sorted form, plain-load form (any length, any 4-byte address, one-row bands) -, the counters' export.
// One translation unit per kernel family (round 5): what is shared lives in lshrs_common.h, measurement switches (-DLSHRS_AB_*,
// tools/ab_build.py) are local to the unit whose kernel they alter and reported through lshrs_build_flags().
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
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

// The library's scalar tail: the dim % 4 elements behind the last group of four are added in plain C behind all blocks
// (lshrs_tb_model_row_dot) - tail_model 1 as OpenBLAS's SkylakeX build contracts that C, 2 as its Haswell / Zen build leaves it.
// pt / xt: the first tail element of the hyperplane / the row; returns y with the tail, adds the tail's x^2 to *ss where asked.
__device__ __forceinline__ float blas_scalar_tail(float y, const float* __restrict__ pt, const float* __restrict__ xt, int m3,
                                                  int tail_model, bool count_ss, float* ss) {
  const float a0 = pt[0], x0 = xt[0];
  const float a1 = m3 > 1 ? pt[1] : 0.f, x1 = m3 > 1 ? xt[1] : 0.f;
  const float a2 = m3 > 2 ? pt[2] : 0.f, x2 = m3 > 2 ? xt[2] : 0.f;
  if (count_ss) *ss = __builtin_fmaf(x0, x0, __builtin_fmaf(x1, x1, __builtin_fmaf(x2, x2, *ss)));
  if (tail_model == 2) {                                     // nothing contracted
    float t = mul_then_add(0.f, a0, x0);
    if (m3 > 1) t = mul_then_add(t, a1, x1);
    if (m3 > 2) t = mul_then_add(t, a2, x2);
    return mul_then_add(y, t, 1.0f);
  }
  if (m3 == 1) return __builtin_fmaf(a0, x0, y);
  float t = __builtin_fmaf(a0, x0, mul_then_add(0.f, a1, x1));
  if (m3 > 2) t = __builtin_fmaf(a2, x2, t);
  return mul_then_add(y, t, 1.0f);
}

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
// SAMEP (REPLAY only; round 5): the list comes BY KEY COLUMN - stage 1 appended every flagged (and sampled) projection to its
// column's segment (BUCKETS: lshrs_sig_sort) -, every group of eight entries shares one hyperplane: its row is fetched ONCE per
// slab (one LDS-DMA of SLAB x 128 bytes by 8 SLAB lanes) and read by all eight entries from the same LDS words, instead of eight
// times from L2 - the x rows are then the only stream (72.6 against 98.7 us per 115 k entries at 768-d,
// profiles/r03_stage2_streams.log).  (Round 5 also had a counting sort of the plain list in front of this instantiation - three
// more launches, slower than the buckets at every shape: gone in round 6.)
template <bool REPLAY, bool GENERAL = false, int SLAB = kFixSlabG, bool SAMEP = false>
__global__ __launch_bounds__(64) void sig_fix8_kernel(const FixArgs a) {
  static_assert(!SAMEP || (REPLAY && SLAB * 8 <= 64), "the shared hyperplane slab is one LDS-DMA of the wave");
  __shared__ __attribute__((aligned(16))) f32x4 xs[2][SLAB * 8 * kFixG];
  __shared__ __attribute__((aligned(16))) f32x4 ps[2][SAMEP ? 64 : SLAB * 8 * kFixG];
  const int lane = threadIdx.x, g = lane & (kFixG - 1), sub = lane >> 3;
  const int shh = sub >> 2, sq = sub & 3;           // this lane's chunk of every k-tile: k = 32 t + 16 shh + 4 sq + 0..3
  const int64_t* __restrict__ list = SAMEP ? a.sorted_list : a.flag_list;
  const float* __restrict__ ylist = SAMEP ? a.sorted_y : a.flag_y;
  // BUCKETS (SAMEP with a.col_cap > 0): stage 1 left every entry in its column's segment; the groups of eight are numbered
  // through the columns - gstart[c] = groups in front of column c, by a scan every workgroup does for itself (<= 1024 counts)
  __shared__ int gstart[SAMEP ? kSortMaxCols + 1 : 1];
  const bool buckets = SAMEP && a.col_cap > 0;
  if (SAMEP && buckets) {
    constexpr int kPer = kSortMaxCols / 64;
    int mine[kPer], sum = 0;
    bool over = false;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int c = lane * kPer + i;
      const int want = c < a.padcols ? a.col_count[c] : 0;
      over = over || want > a.col_cap;
      mine[i] = ((want < a.col_cap ? want : a.col_cap) + kFixG - 1) / kFixG;
      sum += mine[i];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    int at = incl - sum;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      gstart[lane * kPer + i] = at;
      at += mine[i];
    }
    if (lane == 63) gstart[kSortMaxCols] = at;
    if (blockIdx.x == 0 && __any(over) && lane == 0) *a.overflow = 1;
    __syncthreads();
  }
  const int cnt = !SAMEP ? min(*a.flag_count, a.flag_cap) : gstart[kSortMaxCols] * kFixG;
  const int fgroups = (cnt + kFixG - 1) / kFixG;
  const int groups = fgroups + (!SAMEP && REPLAY && a.audit_list != nullptr ? (a.audit_n + kFixG - 1) / kFixG : 0);   // audit groups behind the list's
  const size_t ldp = (size_t)a.ktiles * kKTile;
  const int head = GENERAL ? (a.dim & 4) : 0;                 // 8 m + 4 elements: the first four go ahead of the tiles
  const int m3 = GENERAL ? (a.dim & 3) : 0;                   // the library's scalar tail behind the last group of four (round 5)
  const int body = GENERAL ? a.dim - m3 - head : a.ktiles * kKTile; // elements the tiles cover (from element `head` on)
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
    if (SAMEP && buckets) {
      int lo = 0, hi = a.padcols - 1;                // the column this group belongs to: the last c with gstart[c] <= grp
      while (lo < hi) {                              // (uniform per wave: ten LDS reads of one address each)
        const int mid = (lo + hi + 1) >> 1;
        if (gstart[mid] <= grp) lo = mid; else hi = mid - 1;
      }
      const int off = (grp - gstart[lo]) * kFixG + g;
      const int have = min(a.col_count[lo], a.col_cap);
      inlist = off < have;
      it.e = lo * a.col_cap + (inlist ? off : 0);    // (a short last group re-does the column's first entry, unused)
      item = list[it.e];
      it.audit = (item & kAuditBit) != 0;            // the sample of un-flagged projections rides in the segments, marked
      item &= ~kAuditBit;
    } else if (!it.audit) {
      it.e = grp * kFixG + g;
      inlist = it.e < cnt;
      item = list[inlist ? it.e : grp * kFixG];
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
      if (GENERAL && m3 != 0)       // dim % 4 elements of scalar tail behind all blocks: every lane of the entry, plain loads
        yb = blas_scalar_tail(yb, a.prow + (size_t)col * ldp + head + body, cur.xrow + head + body, m3, a.tail_model, sub == 0, &ss);
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
        const float y1 = (SAMEP && buckets) ? ylist[e] : a.audit_vals[2 * e];
        const float thr = (SAMEP && buckets) ? a.flag_thr[e] : a.audit_vals[2 * e + 1];
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
// blas model 3 (lshrs_host.h, tb_model_small_skx): OpenBLAS's SkylakeX build on bands of two rows and more over at most eight
// elements - rows in blocks of 16 / 8 / 4 / 2 / 1, each with its own arithmetic per length.  One lane computes the whole value.
__device__ __forceinline__ float mul_rounded(float a, float b) {      // (never contracted into an fma with the addition behind it:
#pragma clang fp contract(off)                                          //  hipcc's __fmul_rn is a plain product, which it would be)
  return a * b;
}
__device__ __forceinline__ float add_rounded(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float small_skx_dot(const float* __restrict__ a, const float* __restrict__ x, int n, int row, int rows) {
  auto p = [&](int k) { return mul_rounded(a[k], x[k]); };
  auto f = [&](int k, float s) { return __builtin_fmaf(a[k], x[k], s); };
  const int b16 = rows / 16 * 16, b8 = rows / 8 * 8;
  int rem = rows - b8, type;
  if (row < b16) type = 16;
  else if (row < b8) type = 8;
  else if (rem >= 4 && row < b8 + 4) type = 4;
  else {
    const int base = b8 + (rem >= 4 ? 4 : 0);
    rem -= rem >= 4 ? 4 : 0;
    type = (rem >= 2 && row < base + 2) ? 2 : 1;
  }
  auto seq = [&]() {
    float s = p(0);
    for (int k = 1; k < n; ++k) s = add_rounded(s, p(k));
    return s;
  };
  auto chain = [&]() {
    float s = p(0);
    for (int k = 1; k < n; ++k) s = f(k, s);
    return s;
  };
  switch (n) {
    case 1: return p(0);
    case 2: return type == 16 ? f(1, p(0)) : add_rounded(p(0), p(1));
    case 3:
      if (type >= 8) return f(2, f(1, p(0)));
      if (type == 2) return add_rounded(add_rounded(p(0), p(1)), p(2));
      return f(2, f(0, p(1)));
    case 4: return type == 1 ? seq() : add_rounded(add_rounded(p(0), p(1)), add_rounded(p(2), p(3)));
    case 5: return type >= 4 ? chain() : seq();
    case 6:
      if (type >= 8) return chain();
      if (type == 4) return add_rounded(add_rounded(p(0), f(1, p(2))), add_rounded(p(3), f(4, p(5))));
      return seq();
    case 7:
      if (type >= 8) return chain();
      if (type == 4) return add_rounded(add_rounded(f(0, p(1)), f(4, p(5))), add_rounded(f(2, p(3)), p(6)));
      return seq();
    default:
      if (type >= 4)
        return add_rounded(add_rounded(add_rounded(p(0), p(1)), add_rounded(p(2), p(3))), add_rounded(add_rounded(p(4), p(5)), add_rounded(p(6), p(7))));
      if (type == 2)
        return add_rounded(add_rounded(add_rounded(p(0), p(1)), add_rounded(p(4), p(5))), add_rounded(add_rounded(p(2), p(3)), add_rounded(p(6), p(7))));
      return add_rounded(add_rounded(add_rounded(add_rounded(p(0), p(4)), add_rounded(p(1), p(5))), add_rounded(p(2), p(6))), add_rounded(p(3), p(7)));
  }
}

// (Round 6: also stage 2 of the SPLIT pass for bands of one row - stage 1 on the matrix cores, the sdot replay here: the stage-1
// value of every list entry is measured against the host's (a.flag_y -> slot [2]) and the audit sample of un-flagged projections
// is verified behind the list (a.audit_list / a.audit_vals -> slots [3] [4] [5]), as sig_fix8_kernel does.)
__global__ __launch_bounds__(64) void sig_fixany_kernel(const FixArgs a) {
  const int lane = threadIdx.x, g = lane & (kFixG - 1), sub = lane >> 3;
  const int cnt = min(*a.flag_count, a.flag_cap);
  const int fgroups = (cnt + kFixG - 1) / kFixG;
  const int groups = fgroups + (a.audit_list != nullptr ? (a.audit_n + kFixG - 1) / kFixG : 0);     // audit groups behind the list's
  const size_t ldp = (size_t)a.ktiles * kKTile;
  const int body = a.dim & ~3, m3 = a.dim & 3;
  int n_ties = 0, n_flips = 0, n_aud = 0, n_abad = 0;
  float max_dev = 0.f, max_ratio = 0.f;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {     // (uniform per wave)
    const bool audit = grp >= fgroups;                              // (uniform per wave: a group is the list's or the audit's)
    const int e = (audit ? grp - fgroups : grp) * kFixG + g;
    int64_t item;
    bool inlist;
    if (!audit) {
      inlist = e < cnt;
      item = a.flag_list[inlist ? e : grp * kFixG];
    } else {
      item = e < a.audit_n ? a.audit_list[e] : -1;
      inlist = item >= 0;
      if (!inlist) item = 0;                                        // (an empty slot: row 0, column 0 - fetched, never used)
    }
    const int64_t row = item >> 21;
    const int col_raw = (int)(item & ((1 << 21) - 1));
    const bool live = inlist && col_raw < a.padcols;
    const int col = col_raw < a.padcols ? col_raw : 0;
    const float* __restrict__ xr = a.X + row * a.ldx;
    const float* __restrict__ pr = a.prow + (size_t)col * ldp;
    const int kind = blas_row_kind(col % a.band_cols, a.rows_per_band);
    float y = 0.f, ss = 0.f;
    if (a.tail_model == 3) {
      // the SkylakeX build's small-matrix kernels (at most eight elements, bands of two rows and more): every lane of the entry
      const int j = col % a.band_cols;
      y = j < a.rows_per_band ? small_skx_dot(pr, xr, a.dim, j, a.rows_per_band) : 0.f;
      if (sub == 0)
        for (int k = 0; k < a.dim; ++k) ss = __builtin_fmaf(xr[k], xr[k], ss);
    } else if (a.rows_per_band == 1) {
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
    if (m3 != 0 && a.rows_per_band != 1 && a.tail_model != 3)      // the scalar tail (lshrs_tb_model_row_dot)
      y = blas_scalar_tail(y, pr + body, xr + body, m3, a.tail_model, sub == 0, &ss);
    float s2 = ss + __shfl(ss, (lane + 32) & 63);
    s2 += __shfl(s2, (lane + 8) & 63);
    s2 += __shfl(s2, (lane + 16) & 63);
    if (sub == 0 && live && audit) {
      // a projection stage 1 decided on its own: its key bit must be the sign of the host's value, its stage-1 value within the
      // window it was compared with; nothing is patched (sig_fix8_kernel's audit)
      const uint8_t kbyte = a.keys[row * (int64_t)a.row_bytes + (col >> 3)];
      ++n_aud;
      if ((y > 0.f) != (((kbyte >> (col & 7)) & 1) != 0)) ++n_abad;
      const float thr = a.audit_vals[2 * e + 1];
      if (thr > 0.f) {
        const float ratio = __builtin_fabsf(a.audit_vals[2 * e] - y) / thr;
        if (ratio < __builtin_inff()) max_ratio = __builtin_fmaxf(max_ratio, ratio);
      }
    } else if (sub == 0 && live) {
      uint8_t* kb = a.keys + row * (int64_t)a.row_bytes + (col >> 3);
      const uintptr_t addr = reinterpret_cast<uintptr_t>(kb);
      unsigned int* w32 = reinterpret_cast<unsigned int*>(addr & ~(uintptr_t)3);
      const unsigned int bitmask = 1u << (8 * (unsigned)(addr & 3) + (col & 7));
      const bool want = y > 0.f;                                   // (0, -0 and NaN give 0: lsh.py:204)
      const bool have = (*kb >> (col & 7)) & 1;
      if (__builtin_fabsf(y) < a.tau * sqrtf(s2) * a.tie_coef[col]) ++n_ties;
      if (a.flag_y != nullptr) {                                   // the live margin of stage 1, in the units its window is given in
        const float scale = sqrtf(s2) * a.norms[col];
        if (scale > 0.f) {
          const float dev = __builtin_fabsf(a.flag_y[e] - y) / (scale * 0x1p-24f);
          if (dev < __builtin_inff()) max_dev = __builtin_fmaxf(max_dev, dev);   // (NaN - a row flagged wholesale - drops out)
        }
      }
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
    max_dev = __builtin_fmaxf(max_dev, __shfl_xor(max_dev, off));
    n_aud += __shfl_xor(n_aud, off);
    n_abad += __shfl_xor(n_abad, off);
    max_ratio = __builtin_fmaxf(max_ratio, __shfl_xor(max_ratio, off));
  }
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

// Behind stage 2 of a replay pass: folds the per-workgroup statistics (nparts slots of 3 ints behind the
// LSHRS_SIG_COUNTERS counters: ties, sign flips, max deviation) into the counters, hands the counters to the host (pinned
// memory) and leaves the whole block zeroed for the next call: one small launch instead of a copy and a fill.
// Without host_counts the folded counters stay in the device block (the caller copies it).
constexpr int kExportThreads = 1024;      // one part or two per thread: the fold is one memory round trip deep, not nparts / 64
__global__ __launch_bounds__(kExportThreads) void export_counts_kernel(int* counters, int* host_counts, int nparts, int* zero = nullptr,
                                                                      int nzero = 0, int* done = nullptr, int epoch = 0) {
  __shared__ int fold[kExportThreads / 64][kFixParts];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < nzero; i += kExportThreads) zero[i] = 0;      // (buckets: the column counters of the NEXT launch)
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
  if (done != nullptr) {        // (lshrs_wait_done: the counters first - and everything the pass did before them -, then the word the host polls)
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
}  // namespace

uint32_t lshrs_flags_replay(void) {
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
#if LSHRS_FIX_SLAB != 6 || LSHRS_FIX_GRID != 1536
  f |= LSHRS_BUILD_TUNED | (1u << 21);
#endif
#if defined(LSHRS_BUCKET_GRID) || defined(LSHRS_BUCKET_SLAB)
  f |= LSHRS_BUILD_TUNED | (1u << 22);
#endif
  return f;
}

// Stage 2 of a split pass behind its stage 1 (sig_split.hip: split_pass), on stream s: the flagged projections of f.flag_list,
// one by one.  blas_model 0: the canonical f32 chain, ties reported in f.tie_list.  > 0: the host BLAS's order replayed, keys
// patched, the audit sample verified, the statistics folded into `counters` and handed to `host_counts`; where the caller
// gave the scratch (lshrs_sig_sort) and the rows are long, column by column through the buckets stage 1 filled.
int lshrs_replay_stage2(const FixArgs& f, int32_t* counters, int32_t* host_counts, const Opts& o, hipStream_t s) {
  const int blas_model = f.blas_model, rows_per_band = f.rows_per_band, dim = f.dim, flag_cap = f.flag_cap;
  const int64_t* flag_list = f.flag_list;
  const float* flag_y = f.flag_y;
  const int* flag_count = f.flag_count;
  const int64_t groups = ((int64_t)flag_cap + kFixG - 1) / kFixG;
  const bool short_rows = f.ktiles <= kFixSlabShort && blas_model != 0;
  const int grid_cap = short_rows ? kFixGridShort : kFixGridG;
  const dim3 grid((unsigned)(groups < grid_cap ? groups : grid_cap)), block(64);
  if (blas_model != 0) {
    int nparts = (int)grid.x;
    if (rows_per_band == 1) {
      // a band of ONE row: the host calls sdot - replayed by the plain-load form, at every length (round 6: behind the matrix
      // cores' stage 1; round 5 had these shapes on the exact-f32 kernel: 0.05 of the HBM roof at 64 x 1 x 100)
      const int64_t agroups = f.audit_list != nullptr ? ((int64_t)f.audit_n + kFixG - 1) / kFixG : 0;
      const int64_t want = groups + agroups;
      const dim3 ogrid((unsigned)(want < kFixGridG ? want : kFixGridG));
      hipExtLaunchKernelGGL(sig_fixany_kernel, ogrid, block, 0, s, o.ev[2], o.ev[3], 0, f);
      nparts = (int)ogrid.x;
    } else if (short_rows) {
      if (blas_general(rows_per_band, f.ktiles, dim))
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, true, kFixSlabShort>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
      else
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, false, kFixSlabShort>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    } else if (f.col_cap > 0) {
      // BUCKETS: stage 1 left every flagged (and sampled) projection in its column's segment - stage 2 with ONE hyperplane per
      // group of eight at once, no sort, no launch of its own for the audit sample; the launch behind it also clears the
      // column counters the next pass will use (the two sets alternate: this pass's stay readable for the live audit)
#ifndef LSHRS_BUCKET_GRID
#define LSHRS_BUCKET_GRID 2048      // eight single-wave workgroups per CU.  A/B builds (profiles/r05_bucket_grid_ab.log), stage 2 at config 2 / config 5 in ms:
                                    // 2048 x 6 tiles 0.088 / 2.47; 1536: 0.102 / 2.79; 2560: 0.115 / 3.32; 2816: 0.105 / 3.20; slabs of 4: 0.094 / 2.75; of 8: 0.121 / 3.70
#endif
#ifndef LSHRS_BUCKET_SLAB
#define LSHRS_BUCKET_SLAB LSHRS_FIX_SLAB
#endif
      constexpr int kBucketGrid = LSHRS_BUCKET_GRID, kBucketSlab = LSHRS_BUCKET_SLAB;
      const int64_t bgroups = (int64_t)f.padcols * ((f.col_cap + kFixG - 1) / kFixG);
      const dim3 bgrid((unsigned)(bgroups < kBucketGrid ? bgroups : kBucketGrid));
      FixArgs fb = f;
      fb.audit_list = nullptr;
      fb.audit_n = 0;
      fb.overflow = counters + 7;
      if (blas_general(rows_per_band, f.ktiles, dim))
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, true, kBucketSlab, true>), bgrid, block, 0, s, o.ev[2], o.ev[3], 0, fb);
      else
        hipExtLaunchKernelGGL((sig_fix8_kernel<true, false, kBucketSlab, true>), bgrid, block, 0, s, o.ev[2], o.ev[3], 0, fb);
      hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kExportThreads), 0, s, counters, host_counts, (int)bgrid.x,
                         o.sort->hist + (size_t)(1 - (o.sort->parity & 1)) * kSortMaxCols, kSortMaxCols, o.done, o.epoch);
      return -(int)hipGetLastError();
    } else if (blas_general(rows_per_band, f.ktiles, dim))
      hipExtLaunchKernelGGL((sig_fix8_kernel<true, true>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    else
      hipExtLaunchKernelGGL((sig_fix8_kernel<true, false>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
    hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kExportThreads), 0, s, counters, host_counts, nparts, nullptr, 0, o.done,
                       o.epoch);
  } else {
    hipExtLaunchKernelGGL((sig_fix8_kernel<false, false>), grid, block, 0, s, o.ev[2], o.ev[3], 0, f);
  }
  return -(int)hipGetLastError();
}

extern "C" {

int lshrs_sig_resolve_ties_replay_f32(const float* X, int64_t n, int64_t ldx, const void* workspace, int32_t num_bands,
                                       int32_t rows_per_band, int32_t dim, uint8_t* keys, const int64_t* tie_list,
                                       int32_t tie_cap, int32_t* counters, float tau, int64_t* flag_list,
                                       int32_t flag_cap, int32_t blas_model, int32_t* host_counts, void* stream) {
  int32_t* tie_count = counters;                          // [0] tie entries the f32 kernel wanted to write
  int32_t* flag_count = counters != nullptr ? counters + 1 : nullptr;   // [1] items expanded from them
  if (n == 0) return 0;
  if (X == nullptr || workspace == nullptr || keys == nullptr || tie_list == nullptr || tie_count == nullptr ||
      flag_list == nullptr || flag_count == nullptr || tie_cap <= 0 || flag_cap <= 0 || n < 0 || ldx < dim ||
      !sig_shape_ok(num_bands, rows_per_band, dim) || (blas_model != 1 && blas_model != 2 && blas_model != 3) ||
      (blas_model == 3 && (dim > 8 || rows_per_band < 2)))     // (model 3: the SkylakeX build's small-matrix kernels only)
    return LSHRS_E_BADARG;
  const SigGeom g = sig_geom(num_bands, rows_per_band, dim);
  const int row_bytes = num_bands * g.bb;
  // stage 2 stages 16-byte chunks of 16-byte aligned rows (key rows may have any width: split_pass's comment); anything
  // else - dim % 4 elements of scalar tail (blas_model 1 / 2: how the host compiles it), rows that are only 4-byte aligned -
  // goes through the plain-load form of the same replay (sig_fixany_kernel)
  const bool fast = dim % 4 == 0 && dim >= 8 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && rows_per_band >= 2 &&
                    blas_model != 3;
  const int body = dim & ~3;
  // (a band of ONE row is sdot on the host: modelled for every length, both builds - the plain-load form follows it)
  const bool one_row = rows_per_band == 1;
  // (fewer than 9 elements with a scalar tail: only the Haswell / Zen build's order - model 2 - is modelled there)
  // (8 m + 4 elements beyond 4096 - a short last block behind full ones - only through the plain-load form, which takes every
  //  block's first four first; the LDS-DMA form takes the ROW's first four first: right up to 4096 elements)
  if (!one_row && dim < 9 && dim % 4 != 0 && blas_model == 1) return LSHRS_E_TOOLARGE;
  const bool short_last_block = body % 8 != 0 && body > 4096;
  if (n >= ((int64_t)1 << 42) || (fast && !short_last_block && blas_model != 1)) return LSHRS_E_TOOLARGE;
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
    if (!fast || short_last_block) hipLaunchKernelGGL(sig_fixany_kernel, grid, block, 0, s, f);
    else if (blas_general(rows_per_band, g.ktiles, dim)) hipLaunchKernelGGL((sig_fix8_kernel<true, true>), grid, block, 0, s, f);
    else hipLaunchKernelGGL((sig_fix8_kernel<true, false>), grid, block, 0, s, f);
    hipLaunchKernelGGL(export_counts_kernel, dim3(1), dim3(kExportThreads), 0, s, counters, host_counts, (int)grid.x);
  }
  return -(int)hipGetLastError();
}

}  // extern "C"
