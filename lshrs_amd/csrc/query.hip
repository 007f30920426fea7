// query.hip - part of liblshrs_hip.so, the gfx950 (MI355X / CDNA4) implementation of the lshrs hot path.
// The candidate path of a batch of queries, between the signature pass and the cosine rerank: what the
// reference does one dict entry at a time in LSHRS._candidate_counts (lshrs/core/main.py:1088-1111) and the sort of
// LSHRS.query (:614), and its top-p / top-k cut (:650-657) - bucket lookup by bisection in the device-resident bucket
// arrays, the members of a query's buckets sorted, counted and ordered inside ONE workgroup's LDS (a bitonic network over
// 64-bit items; the lists are a few hundred to a few thousand entries: they never go through global memory), the
// per-query order by score, and the compaction of what the caller asked for.  Integer / index work: bit-exact by
// construction; HBM traffic is the bucket members once and the candidates once.
// ABI and reference citations: include/lshrs_hip.h.  Design notes: DESIGN.md §10.
#include "lshrs_common.h"

using namespace lshrs;

namespace {
constexpr int kQThreads = 256;
constexpr int kQWaves = kQThreads / 64;
constexpr uint64_t kTopBit = 1ull << 63;

struct Segment {   // == lshrs_bucket_segment
  const int64_t* codes;
  const int64_t* offsets;
  const int64_t* members;
  int64_t n_codes;
  const int32_t* directory;   // optional: directory[c] = first bucket whose code is >= c, c = 0 .. dir_codes (dense small code spaces)
  int64_t dir_codes;
};
static_assert(sizeof(Segment) == sizeof(lshrs_bucket_segment), "segment descriptor layout");

__device__ __forceinline__ int pow2_ceil(int v) {
  return v <= 2 ? 2 : 1 << (32 - __builtin_clz((unsigned)(v - 1)));
}

// The same network for up to E x kQThreads items held in REGISTERS (item e * kQThreads + tid is thread tid's e-th): exchanges over
// fewer than 64 positions are lane shuffles, over 64 and 128 positions trips through `items`, over 256 and more the thread's own
// registers.  The list of ONE query is a few hundred items (sixteen buckets of fifteen members at 1 M stored ids): through LDS
// with a workgroup barrier per step its two sorts were 15 of the one-query kernel's 29 us (tools/one_query_stages.py).
template <int E>
__device__ __forceinline__ void bitonic_sort_regs(uint64_t* items, int P) {
  const int tid = threadIdx.x;
  uint64_t v[E];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = e * kQThreads + tid < P ? items[e * kQThreads + tid] : ~0ull;
  __syncthreads();
#pragma unroll
  for (int size = 2; size <= E * kQThreads; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= kQThreads) {                       // both items are this thread's
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int f = e ^ (stride / kQThreads);
          if (f > e) {
            const bool up = (((e * kQThreads + tid) & size) == 0);
            const uint64_t lo = v[e] < v[f] ? v[e] : v[f], hi = v[e] < v[f] ? v[f] : v[e];
            v[e] = up ? lo : hi;
            v[f] = up ? hi : lo;
          }
        }
      } else {
        uint64_t o[E];
        if (stride >= 64) {                            // another wave's: through LDS
#pragma unroll
          for (int e = 0; e < E; ++e) items[e * kQThreads + tid] = v[e];
          __syncthreads();
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = items[e * kQThreads + (tid ^ stride)];
          __syncthreads();
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) o[e] = __shfl_xor((unsigned long long)v[e], stride);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = e * kQThreads + tid;
          const bool keep_min = ((i & size) == 0) == ((i & stride) == 0);
          const uint64_t lo = v[e] < o[e] ? v[e] : o[e], hi = v[e] < o[e] ? o[e] : v[e];
          v[e] = keep_min ? lo : hi;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (e * kQThreads + tid < P) items[e * kQThreads + tid] = v[e];
  __syncthreads();
}

// Ascending bitonic network over P (a power of two) 64-bit items in LDS, all kQThreads threads of the workgroup; the items
// are final - and visible to every thread - when it returns.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* items, int P) {
#ifndef LSHRS_AB_QUERY_LDS_SORT
  if (P <= kQThreads) return bitonic_sort_regs<1>(items, P);
  if (P <= 2 * kQThreads) return bitonic_sort_regs<2>(items, P);
#endif
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += kQThreads) {
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
}

// ------------------------------------------------------------------------------------------
// Bucket lookup: one workgroup per query, one thread per (band, segment) slot.  code = band << 8 B | little-endian key
// (what BucketCSR.codes holds, lshrs_amd/packed_ops.py); the bucket's members are segment.members[start .. start + len).
// Also the running offset of every slot inside the query's pair list and the list's length.
// ------------------------------------------------------------------------------------------
// (the body: also the first phase of query_one_kernel.)  Returns the list's length in every thread.
__device__ __forceinline__ long long lookup_body(const uint8_t* __restrict__ k, int nb, int bb, const Segment* __restrict__ segs,
                                                 int nseg, int64_t* __restrict__ slot_start, int32_t* __restrict__ slot_len,
                                                 int32_t* __restrict__ slot_off, long long* scan) {
  const int tid = threadIdx.x;
  const int nslots = nb * nseg;
  long long carry = 0;
  for (int tile = 0; tile < nslots; tile += kQThreads) {
    const int s = tile + tid;
    long long len = 0;
    int64_t start = 0;
    if (s < nslots) {
      const int b = s / nseg, g = s - b * nseg;
      int64_t code = (int64_t)b << (8 * bb);
      for (int j = 0; j < bb; ++j) code |= (int64_t)k[b * bb + j] << (8 * j);
      const Segment sg = segs[g];
      if (sg.directory != nullptr && code < sg.dir_codes) {
        // a code space small enough to list (keys of 1 or 2 bytes): the bucket's position from a table - two round trips
        // instead of the twenty of a bisection in a million codes (latency is all this kernel has)
        const int lo = sg.directory[code];
        if (lo < sg.directory[code + 1]) {
          start = sg.offsets[lo];
          len = sg.offsets[lo + 1] - start;
        }
      } else {
        int64_t lo = 0, hi = sg.n_codes;
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (sg.codes[mid] < code) lo = mid + 1; else hi = mid;
        }
        if (lo < sg.n_codes && sg.codes[lo] == code) {
          start = sg.offsets[lo];
          len = sg.offsets[lo + 1] - start;
        }
      }
      slot_start[s] = start;
      slot_len[s] = (int32_t)(len > 0x7fffffffLL ? 0x7fffffffLL : len);
    }
    // inclusive scan of the tile's lengths, then exclusive + carry
    scan[tid] = len;
    __syncthreads();
    for (int off = 1; off < kQThreads; off <<= 1) {
      const long long v = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    if (s < nslots) {
      const long long ex = carry + scan[tid] - len;
      slot_off[s] = (int32_t)(ex > 0x7fffffffLL ? 0x7fffffffLL : ex);
    }
    carry += scan[kQThreads - 1];
    __syncthreads();
  }
  return carry;
}

__global__ __launch_bounds__(kQThreads) void query_lookup_kernel(const uint8_t* __restrict__ keys, int nb, int bb,
                                                                 const Segment* __restrict__ segs, int nseg,
                                                                 int64_t* __restrict__ slot_start,
                                                                 int32_t* __restrict__ slot_len,
                                                                 int32_t* __restrict__ slot_off,
                                                                 int32_t* __restrict__ pair_count) {
  __shared__ long long scan[kQThreads];
  const int qi = blockIdx.x;
  const int64_t so = (int64_t)qi * nb * nseg;
  const long long total = lookup_body(keys + (int64_t)qi * nb * bb, nb, bb, segs, nseg, slot_start + so, slot_len + so,
                                      slot_off + so, scan);
  if (threadIdx.x == 0) pair_count[qi] = (int32_t)(total > 0x7fffffffLL ? 0x7fffffffLL : total);
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of q small counts by ONE workgroup (q is a batch of queries: thousands, not billions), the sum and the
// maximum beside it.  With `keep_out` the counts are first cut to what the caller asked for - the top-p / top-k limit of
// LSHRS.query (lshrs/core/main.py:650-657): max(1, ceil(n * top_p)) in the double arithmetic `math.ceil(len(scored) * top_p)`
// runs in, then min(., top_k).
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

__device__ __forceinline__ int keep_of(int u, int top_k, double top_p) {
  if (u <= 0) return 0;
  long long k = u;
  if (top_p >= 0.0) {
    const double lim = ceil((double)u * top_p);
    k = lim < 1.0 ? 1 : (lim > (double)u ? u : (long long)lim);
  }
  if (top_k >= 0 && k > top_k) k = top_k;
  return (int)k;
}

__global__ __launch_bounds__(kScanThreads) void query_scan_kernel(const int32_t* __restrict__ counts, int q, int top_k,
                                                                  double top_p, int32_t* __restrict__ keep_out,
                                                                  int64_t* __restrict__ offsets,
                                                                  int64_t* __restrict__ totals) {
  __shared__ long long part[kScanThreads];
  __shared__ int pmax[kScanThreads];
  const int tid = threadIdx.x;
  const int per = (q + kScanThreads - 1) / kScanThreads;
  const int lo = tid * per, hi = min(q, lo + per);
  long long sum = 0;
  int mx = 0;
  for (int i = lo; i < hi; ++i) {
    const int v = keep_out != nullptr ? keep_of(counts[i], top_k, top_p) : counts[i];
    sum += v;
    mx = max(mx, v);
  }
  part[tid] = sum;
  pmax[tid] = mx;
  __syncthreads();
  for (int off = 1; off < kScanThreads; off <<= 1) {
    const long long v = tid >= off ? part[tid - off] : 0;
    const int m = tid >= off ? pmax[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    pmax[tid] = max(pmax[tid], m);
    __syncthreads();
  }
  long long run = part[tid] - sum;
  for (int i = lo; i < hi; ++i) {
    const int v = keep_out != nullptr ? keep_of(counts[i], top_k, top_p) : counts[i];
    if (keep_out != nullptr) keep_out[i] = v;
    offsets[i] = run;
    run += v;
  }
  if (tid == kScanThreads - 1) {
    offsets[q] = part[tid];
    if (totals != nullptr) {
      totals[0] = part[tid];
      totals[1] = pmax[tid];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Collision counting + candidate order of one query inside one workgroup's LDS.
//   items   (member << bbits) | band for every member of every bucket the query's keys select (bit 63 free: the caller
//           has checked member < 2^(63 - bbits))
//   sort    equal (member, band) pairs become neighbours - an id that reached one bucket through two sources (indexed by
//           two calls: two array segments) counts ONCE per band, buckets are sets (lshrs/storage/redis.py:408-416 SADD) -
//           and are struck out; the members' runs are then exactly their collision counts (lshrs/core/main.py:1101-1109)
//   order   key = (bands - count) << (63 - bbits) | member, sorted ascending == sorted by (-count, id)  (main.py:614)
// SRC 0: members fetched from the bucket segments through the lookup's slots; SRC 1: pairs handed in by the host (a storage
// that only has get_bucket - Redis: main.py:1103).
// ------------------------------------------------------------------------------------------
// (the body: items loaded -> the candidates, ordered, at cand_ids[0 .. U) (+ their collisions); also the second phase of
//  query_one_kernel.)  `so`: this query's slots.  Returns U in every thread.
#ifdef LSHRS_AB_ONE_PROBE      // (A/B builds only, tools/one_query_stages.py: where ONE query's kernel spends its time - s_memrealtime, 10 ns ticks)
__device__ unsigned long long g_probe[16];
#define LSHRS_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_probe[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSHRS_PROBE(i) do { } while (0)
#endif

template <int SRC>
__device__ __forceinline__ int collide_body(uint64_t* items, int L, const Segment* __restrict__ segs, int nseg, int nb, int bbits,
                                            const int64_t* __restrict__ slot_start, const int32_t* __restrict__ slot_len,
                                            const int32_t* __restrict__ slot_off, const int64_t* __restrict__ pair_members,
                                            const int32_t* __restrict__ pair_bands, int64_t* __restrict__ cand_ids,
                                            int32_t* __restrict__ cand_hits, int* n_dup, int* n_head) {
  const int tid = threadIdx.x;
  const int P = pow2_ceil(L);
  if (tid == 0) { *n_dup = 0; *n_head = 0; }
  if (SRC == 0) {
    // one THREAD per pair: its slot is the last one whose offset is <= t (a bisection in the slots' running offsets - empty
    // slots share their offset with the slot behind them, so the last one at or below t is the one that holds t), then one
    // load of the member.  (A wave per bucket walked the slots one after the other: 32 dependent round trips per wave at
    // 16 bands x 8 segments - half of this kernel's time, and most of a single query's.)
    const int nslots = nb * nseg;
    for (int t = tid; t < L; t += kQThreads) {
      int lo = 0, hi = nslots;                              // first slot with offset > t
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (slot_off[mid] <= t) lo = mid + 1; else hi = mid;
      }
      const int sl = lo - 1;
      const int b = sl / nseg, g = sl - b * nseg;
      const int64_t m = segs[g].members[slot_start[sl] + (t - slot_off[sl])];
      items[t] = ((uint64_t)m << bbits) | (uint64_t)b;
    }
  } else {
    for (int t = tid; t < L; t += kQThreads) items[t] = ((uint64_t)pair_members[t] << bbits) | (uint64_t)pair_bands[t];
  }
  for (int t = L + tid; t < P; t += kQThreads) items[t] = ~0ull;
  LSHRS_PROBE(3);
  bitonic_sort_lds(items, P);
  LSHRS_PROBE(4);

  // an id twice in one bucket (two sources): strike the repeats, close the gaps with one more sort (rare: skipped when none)
  uint64_t dupmask = 0;        // (P / 256 <= 64 items per thread: one bit each)
  for (int t = tid, j = 0; t < L; t += kQThreads, ++j)
    if (t > 0 && items[t] == items[t - 1]) dupmask |= 1ull << j;
  if (dupmask) atomicAdd(n_dup, __builtin_popcountll(dupmask));
  __syncthreads();
  const int dups = *n_dup;
  if (dups) {
    for (int t = tid, j = 0; t < L; t += kQThreads, ++j)
      if ((dupmask >> j) & 1) items[t] = ~0ull;
    bitonic_sort_lds(items, P);
    L -= dups;
  }
  // heads of the members' runs: marked in bit 63, then each head walks to the next one (a run is at most `bands` long)
  const uint64_t low = ((uint64_t)1 << bbits) - 1;
  uint64_t headmask = 0;
  for (int t = tid, j = 0; t < L; t += kQThreads, ++j)
    if (t == 0 || (items[t] >> bbits) != (items[t - 1] >> bbits)) headmask |= 1ull << j;
  __syncthreads();
  for (int t = tid, j = 0; t < L; t += kQThreads, ++j)
    if ((headmask >> j) & 1) items[t] |= kTopBit;
  if (headmask) atomicAdd(n_head, __builtin_popcountll(headmask));
  __syncthreads();
  // (the band field has done its work: a head keeps its run length - 1 there, read back below by the same thread)
  for (int t = tid, j = 0; t < L; t += kQThreads, ++j)
    if ((headmask >> j) & 1) {
      int e = t + 1;
      while (e < L && !(items[e] & kTopBit)) ++e;
      items[t] = (items[t] & ~low) | (uint64_t)(e - t - 1);
    }
  __syncthreads();
  const int mshift = 63 - bbits;
  for (int t = tid, j = 0; t < L; t += kQThreads, ++j) {
    const uint64_t it = items[t];
    uint64_t key = ~0ull;
    if ((headmask >> j) & 1) {
      const uint64_t cnt = (it & low) + 1;
      key = ((uint64_t)(nb - cnt) << mshift) | ((it & ~kTopBit) >> bbits);
    }
    items[t] = key;
  }
  LSHRS_PROBE(5);
  bitonic_sort_lds(items, P);
  LSHRS_PROBE(6);
  const int U = *n_head;
  const uint64_t mmask = ((uint64_t)1 << mshift) - 1;
  for (int t = tid; t < U; t += kQThreads) {
    const uint64_t key = items[t];
    cand_ids[t] = (int64_t)(key & mmask);
    if (cand_hits != nullptr) cand_hits[t] = nb - (int)(key >> mshift);
  }
  return U;
}

template <int SRC>
__global__ __launch_bounds__(kQThreads) void query_collide_kernel(const Segment* __restrict__ segs, int nseg, int nb,
                                                                  int bbits, const int64_t* __restrict__ slot_start,
                                                                  const int32_t* __restrict__ slot_len,
                                                                  const int32_t* __restrict__ slot_off,
                                                                  const int64_t* __restrict__ pair_members,
                                                                  const int32_t* __restrict__ pair_bands,
                                                                  const int64_t* __restrict__ pair_off,
                                                                  int64_t* __restrict__ cand_ids,
                                                                  int32_t* __restrict__ cand_hits,
                                                                  int32_t* __restrict__ ucount, int max_items) {
  extern __shared__ __attribute__((aligned(16))) uint64_t items[];
  __shared__ int n_dup, n_head;
  const int qi = blockIdx.x;
  const int64_t base = pair_off[qi];
  const int64_t L64 = pair_off[qi + 1] - base;
  if (L64 <= 0 || L64 > max_items) {      // nothing to count - or more than the LDS network was sized for (the caller told us its
    if (threadIdx.x == 0) ucount[qi] = L64 <= 0 ? 0 : -1;    // maximum, or a fixed capacity): -1 says so
    return;
  }
  const int64_t so = (int64_t)qi * nb * nseg;
  const int U = collide_body<SRC>(items, (int)L64, segs, nseg, nb, bbits, SRC == 0 ? slot_start + so : nullptr,
                                  SRC == 0 ? slot_len + so : nullptr, SRC == 0 ? slot_off + so : nullptr,
                                  SRC == 1 ? pair_members + base : nullptr, SRC == 1 ? pair_bands + base : nullptr, cand_ids + base,
                                  cand_hits != nullptr ? cand_hits + base : nullptr, &n_dup, &n_head);
  if (threadIdx.x == 0) ucount[qi] = U;
}

// ------------------------------------------------------------------------------------------
// ONE query - LSHRS.get_top_k / get_above_p, the reference's own calling pattern (lshrs/core/main.py:524-658) - in ONE
// workgroup and ONE launch behind the one-launch signature kernel: lookup, pair list, collision count and order, the cut; the
// ids of a top-k-by-collisions answer go straight into the caller's (pinned) result array and `epoch` into *done behind them;
// a rerank follows in two more launches (cosine_kernel over the candidates left here, query_rank_kernel, which publishes).
// A list beyond max_items: ucount = -1, nothing kept - the host counts.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kQThreads) void query_one_kernel(const uint8_t* __restrict__ keys, int nb, int bb, int bbits,
                                                              const Segment* __restrict__ segs, int nseg,
                                                              int64_t* __restrict__ slot_start, int32_t* __restrict__ slot_len,
                                                              int32_t* __restrict__ slot_off, int max_items, int top_k,
                                                              double top_p, int emit, int64_t* __restrict__ pair_off,
                                                              int64_t* __restrict__ cand_ids, int32_t* __restrict__ ucount,
                                                              int32_t* __restrict__ keep, int64_t* __restrict__ out_off,
                                                              int64_t* __restrict__ out_ids, int* __restrict__ done, int epoch,
                                                              const float* __restrict__ copy_src, float* __restrict__ copy_dst,
                                                              int copy_n) {
  extern __shared__ __attribute__((aligned(16))) uint64_t items[];
  __shared__ long long scan[kQThreads];
  __shared__ int n_dup, n_head;
  const int tid = threadIdx.x;
  // the query vector from pinned host memory into device memory, once, for the rerank launch behind this one (128 workgroups
  // reading it over the link each cost that launch 25 us)
  LSHRS_PROBE(0);
  for (int k = tid; k < copy_n; k += kQThreads) copy_dst[k] = copy_src[k];
#ifdef LSHRS_AB_ONE_PROBE
  __syncthreads();
#endif
  LSHRS_PROBE(1);
  const long long total = lookup_body(keys, nb, bb, segs, nseg, slot_start, slot_len, slot_off, scan);
  LSHRS_PROBE(2);
  int U = 0;
  if (total > max_items) U = -1;
  else if (total > 0) {
    __threadfence_block();          // (the slots were written by other threads of this workgroup: through global memory)
    __syncthreads();
    U = collide_body<0>(items, (int)total, segs, nseg, nb, bbits, slot_start, slot_len, slot_off, nullptr, nullptr, cand_ids,
                        nullptr, &n_dup, &n_head);
  }
  const int K = keep_of(U, top_k, top_p);
  if (tid == 0) {
    pair_off[0] = 0;
    pair_off[1] = total > max_items ? 0 : total;
    ucount[0] = U;
    keep[0] = K;
    out_off[0] = 0;
    out_off[1] = K;
    out_off[2] = U;             // (for the host: candidates found, -1 = beyond the capacity - out_off is where it looks)
  }
  LSHRS_PROBE(7);
  if (emit || K == 0) {            // the answer by collisions (or nothing to rerank): out, and the word the host polls
    __syncthreads();               // (cand_ids[0 .. U) written by other threads above)
    for (int t = tid; t < K; t += kQThreads) out_ids[t] = cand_ids[t];
    if (done != nullptr) {
      __threadfence_system();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  LSHRS_PROBE(8);
}

// ------------------------------------------------------------------------------------------
// Per query: the candidates in descending score (ties by ascending candidate position, NaN last - as lshrs_topk_desc_f32),
// the first keep[q] of them written to the caller's compact result arrays.  scores == NULL: the order the candidates
// already have (collision order: the top_k-by-collisions answer of LSHRS.query, main.py:619-625).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t desc_key(float f) {
  if (f != f) return 0xFFFFFFFFu;
  uint32_t u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}

__global__ __launch_bounds__(kQThreads) void query_rank_kernel(const int64_t* __restrict__ cand_ids,
                                                               const float* __restrict__ scores,
                                                               const int64_t* __restrict__ pair_off,
                                                               const int32_t* __restrict__ ucount,
                                                               const int32_t* __restrict__ keep,
                                                               const int64_t* __restrict__ out_off,
                                                               int64_t* __restrict__ out_ids,
                                                               float* __restrict__ out_scores, int* __restrict__ done,
                                                               int epoch, int max_items) {
  extern __shared__ __attribute__((aligned(16))) uint64_t items[];
  const int tid = threadIdx.x;
  const int qi = blockIdx.x;
  const int K = keep[qi];
  if (K > 0) {                                        // (workgroup-uniform)
    const int64_t base = pair_off[qi], ob = out_off[qi];
    if (scores == nullptr) {
      for (int t = tid; t < K; t += kQThreads) out_ids[ob + t] = cand_ids[base + t];
    } else if (ucount[qi] <= max_items) {        // (a longer list: ranked through global memory by the caller, lshrs_topk_desc_f32)
      const int U = ucount[qi];
      const int P = pow2_ceil(U);
      for (int t = tid; t < P; t += kQThreads)
        items[t] = t < U ? (((uint64_t)desc_key(scores[base + t]) << 32) | (uint32_t)t) : ~0ull;
      bitonic_sort_lds(items, P);
      for (int t = tid; t < K; t += kQThreads) {
        const uint32_t pos = (uint32_t)items[t];
        out_ids[ob + t] = cand_ids[base + pos];
        out_scores[ob + t] = scores[base + pos];
      }
    }
  }
  if (done != nullptr) {     // ONE query answered straight into pinned host memory: its results first, then the word the host polls
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------
// A query whose buckets hold more members than the LDS network takes (LSHRS_QUERY_MAX_PAIRS): the same three steps through
// global memory, one query at a time (rare: 16-bit keys over tens of millions of stored ids).  gather: items[t] as in
// collide_body, padded to `cpad` with ~0; sort (K3's global network); mark: every head of a member's run counts the DISTINCT
// (member, band) pairs of its run (an id twice in one bucket counts once) and becomes (bands - count) << (63 - bbits) | member,
// everything else ~0; sort; emit.
// ------------------------------------------------------------------------------------------
__global__ void query_big_gather_kernel(const Segment* __restrict__ segs, int nseg, int nb, int bbits,
                                        const int64_t* __restrict__ slot_start, const int32_t* __restrict__ slot_off,
                                        int64_t L, int64_t cpad, uint64_t* __restrict__ items) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpad) return;
  if (t >= L) { items[t] = ~0ull; return; }
  const int nslots = nb * nseg;
  int lo = 0, hi = nslots;                                  // first slot with offset > t
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (slot_off[mid] <= t) lo = mid + 1; else hi = mid;
  }
  const int sl = lo - 1;
  const int b = sl / nseg, g = sl - b * nseg;
  const int64_t m = segs[g].members[slot_start[sl] + (t - slot_off[sl])];
  items[t] = ((uint64_t)m << bbits) | (uint64_t)b;
}

__global__ void query_big_mark_kernel(const uint64_t* __restrict__ items, int64_t L, int64_t cpad, int nb, int bbits,
                                      uint64_t* __restrict__ keyed, int* __restrict__ n_head) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cpad) return;
  uint64_t key = ~0ull;
  if (t < L) {
    const uint64_t it = items[t];
    const uint64_t member = it >> bbits;
    if (t == 0 || (items[t - 1] >> bbits) != member) {      // head of the member's run
      int cnt = 1;
      uint64_t prev = it;
      for (int64_t e = t + 1; e < L; ++e) {                 // (a run is at most bands x sources long)
        const uint64_t nx = items[e];
        if ((nx >> bbits) != member) break;
        cnt += nx != prev;
        prev = nx;
      }
      key = ((uint64_t)(nb - cnt) << (63 - bbits)) | member;
      atomicAdd(n_head, 1);
    }
  }
  keyed[t] = key;
}

__global__ void query_big_emit_kernel(const uint64_t* __restrict__ keyed, const int* __restrict__ n_head, int nb, int bbits,
                                      int64_t* __restrict__ cand_ids, int32_t* __restrict__ cand_hits,
                                      int32_t* __restrict__ ucount) {
  const int U = *n_head;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) *ucount = U;
  if (t >= U) return;
  const int mshift = 63 - bbits;
  const uint64_t key = keyed[t];
  cand_ids[t] = (int64_t)(key & (((uint64_t)1 << mshift) - 1));
  if (cand_hits != nullptr) cand_hits[t] = nb - (int)(key >> mshift);
}

int set_lds(const void* fn, size_t bytes) {
  if (bytes <= 48 * 1024) return 0;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : -(int)e;
}
inline size_t items_bytes(int max_items) {
  size_t p = 2 * kQThreads;          // (bitonic_sort_regs exchanges through the first 2 x kQThreads items whatever the list's length)
  while (p < (size_t)max_items) p <<= 1;
  return p * sizeof(uint64_t);
}
}  // namespace

uint32_t lshrs_flags_query(void) {
  uint32_t f = 0;
#ifdef LSHRS_AB_ONE_PROBE
  f |= LSHRS_BUILD_TUNED | (1u << 23);
#endif
#ifdef LSHRS_AB_QUERY_LDS_SORT
  f |= LSHRS_BUILD_TUNED | (1u << 24);
#endif
  return f;
}

extern "C" {

int lshrs_query_lookup_u8(const uint8_t* keys, int32_t q, int32_t num_bands, int32_t band_bytes,
                          const lshrs_bucket_segment* segments, int32_t nseg, int64_t* slot_start, int32_t* slot_len,
                          int32_t* slot_off, int32_t* pair_count, void* stream) {
  if (q == 0) return 0;
  if (keys == nullptr || pair_count == nullptr || q < 0 || num_bands <= 0 || band_bytes <= 0 || nseg < 0) return LSHRS_E_BADARG;
  if (nseg > 0 && (segments == nullptr || slot_start == nullptr || slot_len == nullptr || slot_off == nullptr))
    return LSHRS_E_BADARG;
  if (band_bytes > 6 || num_bands > 32768 || (int64_t)num_bands * nseg > (1 << 24)) return LSHRS_E_TOOLARGE;
  hipLaunchKernelGGL(query_lookup_kernel, dim3((unsigned)q), dim3(kQThreads), 0, static_cast<hipStream_t>(stream), keys,
                     num_bands, band_bytes, reinterpret_cast<const Segment*>(segments), nseg, slot_start, slot_len,
                     slot_off, pair_count);
  return -(int)hipGetLastError();
}

int lshrs_query_scan_i32(const int32_t* counts, int32_t q, int32_t top_k, double top_p, int32_t* keep_out,
                         int64_t* offsets, int64_t* totals, void* stream) {
  if (counts == nullptr || offsets == nullptr || q < 0) return LSHRS_E_BADARG;
  if (top_p > 1.0) return LSHRS_E_BADARG;
  hipLaunchKernelGGL(query_scan_kernel, dim3(1), dim3(kScanThreads), 0, static_cast<hipStream_t>(stream), counts, q,
                     top_k, top_p, keep_out, offsets, totals);
  return -(int)hipGetLastError();
}

static int collide_bits(int32_t num_bands) {
  int bbits = 0;
  while ((1 << bbits) < num_bands) ++bbits;
  return bbits;
}

int lshrs_query_collide_index_i64(const lshrs_bucket_segment* segments, int32_t nseg, int32_t num_bands,
                                  const int64_t* slot_start, const int32_t* slot_len, const int32_t* slot_off,
                                  const int64_t* pair_off, int32_t q, int32_t max_pairs, int64_t* cand_ids,
                                  int32_t* cand_hits, int32_t* ucount, void* stream) {
  if (q == 0) return 0;
  if (pair_off == nullptr || ucount == nullptr || q < 0 || num_bands <= 0 || nseg < 0 || max_pairs < 0) return LSHRS_E_BADARG;
  if (max_pairs > 0 && nseg > 0 && (segments == nullptr || slot_start == nullptr || slot_len == nullptr || slot_off == nullptr ||
                                    cand_ids == nullptr))
    return LSHRS_E_BADARG;
  if (max_pairs > LSHRS_QUERY_MAX_PAIRS || num_bands > 32768) return LSHRS_E_TOOLARGE;
  const size_t shmem = items_bytes(max_pairs);
  const int rc = set_lds(reinterpret_cast<const void*>(query_collide_kernel<0>), shmem);
  if (rc) return rc;
  hipLaunchKernelGGL(query_collide_kernel<0>, dim3((unsigned)q), dim3(kQThreads), shmem, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const Segment*>(segments), nseg, num_bands, collide_bits(num_bands), slot_start,
                     slot_len, slot_off, nullptr, nullptr, pair_off, cand_ids, cand_hits, ucount, (int)(shmem / sizeof(uint64_t)));
  return -(int)hipGetLastError();
}

int lshrs_query_collide_pairs_i64(const int64_t* members, const int32_t* bands, const int64_t* pair_off, int32_t q,
                                  int32_t max_pairs, int32_t num_bands, int64_t* cand_ids, int32_t* cand_hits,
                                  int32_t* ucount, void* stream) {
  if (q == 0) return 0;
  if (pair_off == nullptr || ucount == nullptr || q < 0 || num_bands <= 0 || max_pairs < 0) return LSHRS_E_BADARG;
  if (max_pairs > 0 && (members == nullptr || bands == nullptr || cand_ids == nullptr)) return LSHRS_E_BADARG;
  if (max_pairs > LSHRS_QUERY_MAX_PAIRS || num_bands > 32768) return LSHRS_E_TOOLARGE;
  const size_t shmem = items_bytes(max_pairs);
  const int rc = set_lds(reinterpret_cast<const void*>(query_collide_kernel<1>), shmem);
  if (rc) return rc;
  hipLaunchKernelGGL(query_collide_kernel<1>, dim3((unsigned)q), dim3(kQThreads), shmem, static_cast<hipStream_t>(stream),
                     nullptr, 0, num_bands, collide_bits(num_bands), nullptr, nullptr, nullptr, members, bands, pair_off,
                     cand_ids, cand_hits, ucount, (int)(shmem / sizeof(uint64_t)));
  return -(int)hipGetLastError();
}

static int64_t big_pad(int64_t pairs) {
  int64_t cpad = 4096;
  while (cpad < pairs) cpad <<= 1;
  return cpad;
}

int64_t lshrs_query_big_workspace_bytes(int64_t pairs) {
  if (pairs < 0 || pairs > ((int64_t)1 << 31)) return LSHRS_E_TOOLARGE;
  return 2 * big_pad(pairs) * (int64_t)sizeof(uint64_t) + 16;
}

int lshrs_query_collide_big_i64(const lshrs_bucket_segment* segments, int32_t nseg, int32_t num_bands, const int64_t* slot_start,
                                const int32_t* slot_off, int64_t pairs, void* workspace, int64_t* cand_ids, int32_t* cand_hits,
                                int32_t* ucount, void* stream) {
  if (segments == nullptr || slot_start == nullptr || slot_off == nullptr || workspace == nullptr || cand_ids == nullptr ||
      ucount == nullptr || num_bands <= 0 || nseg <= 0 || pairs <= 0 || (reinterpret_cast<uintptr_t>(workspace) & 7))
    return LSHRS_E_BADARG;
  if (pairs > ((int64_t)1 << 31) - 1 || num_bands > 32768) return LSHRS_E_TOOLARGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t cpad = big_pad(pairs);
  uint64_t* items = static_cast<uint64_t*>(workspace);
  uint64_t* keyed = items + cpad;
  int* n_head = reinterpret_cast<int*>(keyed + cpad);
  const int bbits = collide_bits(num_bands);
  const dim3 grid((unsigned)((cpad + 255) / 256)), block(256);
  hipError_t e = hipMemsetAsync(n_head, 0, sizeof(int), s);
  if (e != hipSuccess) return -(int)e;
  hipLaunchKernelGGL(query_big_gather_kernel, grid, block, 0, s, reinterpret_cast<const Segment*>(segments), nseg, num_bands, bbits,
                     slot_start, slot_off, pairs, cpad, items);
  int rc = lshrs_sort_u64_rows(items, 1, cpad, s);
  if (rc) return rc;
  hipLaunchKernelGGL(query_big_mark_kernel, grid, block, 0, s, items, pairs, cpad, num_bands, bbits, keyed, n_head);
  rc = lshrs_sort_u64_rows(keyed, 1, cpad, s);
  if (rc) return rc;
  hipLaunchKernelGGL(query_big_emit_kernel, grid, block, 0, s, keyed, n_head, num_bands, bbits, cand_ids, cand_hits, ucount);
  return -(int)hipGetLastError();
}

int lshrs_query_one_u8(const uint8_t* keys, int32_t num_bands, int32_t band_bytes, const lshrs_bucket_segment* segments,
                       int32_t nseg, int64_t* slot_start, int32_t* slot_len, int32_t* slot_off, int32_t max_pairs, int32_t top_k,
                       double top_p, int32_t rerank_follows, int64_t* pair_off, int64_t* cand_ids, int32_t* ucount, int32_t* keep,
                       int64_t* out_off, int64_t* out_ids, int32_t* done_host, int32_t epoch, const float* copy_src,
                       float* copy_dst, int32_t copy_n, void* stream) {
  if (copy_n < 0 || (copy_n > 0 && (copy_src == nullptr || copy_dst == nullptr))) return LSHRS_E_BADARG;
  if (keys == nullptr || pair_off == nullptr || cand_ids == nullptr || ucount == nullptr || keep == nullptr || out_off == nullptr ||
      out_ids == nullptr || num_bands <= 0 || band_bytes <= 0 || nseg < 0 || max_pairs <= 0 || top_p > 1.0)
    return LSHRS_E_BADARG;
  if (nseg > 0 && (segments == nullptr || slot_start == nullptr || slot_len == nullptr || slot_off == nullptr)) return LSHRS_E_BADARG;
  if (band_bytes > 6 || num_bands > 32768 || max_pairs > LSHRS_QUERY_MAX_PAIRS || (int64_t)num_bands * nseg > (1 << 24))
    return LSHRS_E_TOOLARGE;
  int* done = nullptr;
  if (done_host != nullptr) {
    void* dp = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dp, done_host, 0);
    if (e != hipSuccess || dp == nullptr) return e != hipSuccess ? -(int)e : LSHRS_E_BADARG;
    done = static_cast<int*>(dp);
  }
  const size_t shmem = items_bytes(max_pairs);
  const int rc = set_lds(reinterpret_cast<const void*>(query_one_kernel), shmem);
  if (rc) return rc;
  hipLaunchKernelGGL(query_one_kernel, dim3(1), dim3(kQThreads), shmem, static_cast<hipStream_t>(stream), keys, num_bands,
                     band_bytes, collide_bits(num_bands), reinterpret_cast<const Segment*>(segments), nseg, slot_start, slot_len,
                     slot_off, (int)(shmem / sizeof(uint64_t)), top_k, top_p, rerank_follows ? 0 : 1, pair_off, cand_ids, ucount,
                     keep, out_off, out_ids, done, (int)epoch, copy_src, copy_dst, (int)copy_n);
  return -(int)hipGetLastError();
}

int lshrs_query_rank_f32(const int64_t* cand_ids, const float* scores, const int64_t* pair_off, const int32_t* ucount,
                         const int32_t* keep, const int64_t* out_off, int32_t q, int32_t max_candidates,
                         int64_t* out_ids, float* out_scores, int32_t* done_host, int32_t epoch, void* stream) {
  if (q == 0) return 0;
  if (cand_ids == nullptr || pair_off == nullptr || ucount == nullptr || keep == nullptr || out_off == nullptr ||
      out_ids == nullptr || q < 0 || max_candidates < 0 || (scores != nullptr && out_scores == nullptr) ||
      (done_host != nullptr && q != 1))
    return LSHRS_E_BADARG;
  if (max_candidates > LSHRS_QUERY_MAX_PAIRS) return LSHRS_E_TOOLARGE;
  int* done = nullptr;
  if (done_host != nullptr) {
    void* dp = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dp, done_host, 0);
    if (e != hipSuccess || dp == nullptr) return e != hipSuccess ? -(int)e : LSHRS_E_BADARG;
    done = static_cast<int*>(dp);
  }
  const size_t shmem = scores != nullptr ? items_bytes(max_candidates) : 0;
  const int rc = set_lds(reinterpret_cast<const void*>(query_rank_kernel), shmem);
  if (rc) return rc;
  hipLaunchKernelGGL(query_rank_kernel, dim3((unsigned)q), dim3(kQThreads), shmem, static_cast<hipStream_t>(stream),
                     cand_ids, scores, pair_off, ucount, keep, out_off, out_ids, out_scores, done, (int)epoch,
                     (int)(shmem / sizeof(uint64_t)));
  return -(int)hipGetLastError();
}

}  // extern "C"

#ifdef LSHRS_AB_ONE_PROBE
extern "C" int lshrs_ab_one_probe(unsigned long long* host16) { return -(int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_probe), 16 * sizeof(unsigned long long)); }
#endif
