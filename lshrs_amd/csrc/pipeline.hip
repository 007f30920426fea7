// pipeline.hip — native driver of the bit-exact signature path for large device-resident batches
// (C ABI: lshrs_pipe_* in include/lshrs_hip.h; design: DESIGN.md "Host pipeline").
//
// Per chunk c of n:
//   caller's stream : signature pass (split-precision or f32 kernel; sig_split.hip, sig_f32.hip)
//   side stream     : c < n-2: export_ties_kernel gathers the tie entries, their X rows and both counters into a
//                     device image, and a speculative prefix of it (the count is not known on the host yet) is copied
//                     to pinned host memory; c >= n-2: the same kernel, on the caller's stream, writes the pinned
//                     memory itself
//   host            : one wait, then resolve() = the reference's own BLAS call on the flagged (row, band) pairs
//                     (lshrs_tb_resolve, liblshrs_host.so); patches land in pinned memory
//   side stream     : scatter_keys_kernel reads the patches from that pinned memory
// The GPU is kept three chunks ahead of the host (a 1M-row batch is four chunks: the host never gates a launch);
// four slots of scratch rotate.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <chrono>
#include <new>

#include "lshrs_hip.h"

namespace {

constexpr int kSlots = 4;
constexpr int kAhead = 3;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Entry e < min(*tie_count, tie_cap): out_entries[e] = tie_list[e], out_rows[e] = X[row of e] (dim floats,
// contiguous); out_counts = (ties wanted, stage-1 entries wanted).  The destinations are a device image or pinned host
// memory (see above).  One WAVE per entry.  Mid-batch the kernel runs beside the next chunk's signature pass, whose
// workgroups take a CU's whole register file: it gets CUs only as those retire, every CU that holds an export wave
// is closed to the pass until that wave is done, and under the pass's HBM stream a wave's dependent loads (entry ->
// row -> store) take several us each - so: enough waves for one or two entries each, not more (measured with host
// destinations: 2 workgroups 200 us per 262 144-row chunk, 16: 90-110 us, 128: as 16 with a shorter tail, 1024:
// signature pass +2 %).
constexpr int kExportBlocks = 128;
template <bool VEC>
__global__ __launch_bounds__(256) void export_ties_kernel(const float* __restrict__ X, int64_t ldx, int dim,
                                                          const int64_t* __restrict__ tie_list,
                                                          const int32_t* __restrict__ counts, int tie_cap,
                                                          int64_t* __restrict__ out_entries,
                                                          float* __restrict__ out_rows,
                                                          int32_t* __restrict__ out_counts) {
  const int wanted = counts[0];
  const int cnt = wanted < tie_cap ? wanted : tie_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_counts[0] = wanted;
    out_counts[1] = counts[1];
  }
  const int lane = threadIdx.x & 63;
  const int waves = gridDim.x * (blockDim.x >> 6);
  for (int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); e < cnt; e += waves) {
    const int64_t e0 = tie_list[2 * (int64_t)e], e1 = tie_list[2 * (int64_t)e + 1];
    if (lane == 0) {
      out_entries[2 * (int64_t)e] = e0;
      out_entries[2 * (int64_t)e + 1] = e1;
    }
    const float* __restrict__ src = X + (e0 >> 16) * ldx;
    float* __restrict__ dst = out_rows + (int64_t)e * dim;
    if (VEC) {
      const int n4 = dim >> 2;
      for (int k0 = 0; k0 < n4; k0 += 256) {        // four 16-byte loads per lane in flight, then their stores
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + 64 * j + lane;
          v[j] = reinterpret_cast<const f32x4*>(src)[k < n4 ? k : 0];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + 64 * j + lane;
          if (k < n4) reinterpret_cast<f32x4*>(dst)[k] = v[j];
        }
      }
    } else {
      for (int k = lane; k < dim; k += 64) dst[k] = src[k];
    }
  }
}

struct Slot {
  int64_t* tie_list = nullptr;     // device
  int64_t* flag_list = nullptr;    // device
  // export image, once in device memory (d_*) and once in pinned host memory (h_*):
  //   head: 64 bytes, int32 (ties wanted, stage-1 entries wanted) | entries int64[tie_cap][2]   (one allocation)
  //   rows: float[tie_cap][dim]
  char* d_head = nullptr;
  float* d_rows = nullptr;
  char* h_head = nullptr;
  int32_t* h_counts = nullptr;     // = h_head
  int64_t* h_entries = nullptr;    // = h_head + 64
  float* h_rows = nullptr;
  int64_t copied = 0;              // entries the speculative copy of this chunk covered
  uint8_t* h_patch = nullptr;      // pinned: rows int64[pairs_cap] | bands int32[pairs_cap] | keys u8[pairs_cap * bb]
  hipEvent_t exported = nullptr, scattered = nullptr;
  // t_end: after the chunk's last kernel on the caller's stream; it releases the export on the side stream.  ONE
  // event packet per chunk boundary (each costs the stream ~6 us of dispatch gap; three per boundary were 5 % of a
  // step).  k1s..k2e: armed only when the caller asked for times - start/stop events riding on the stage-1 and
  // stage-2 dispatches of the split pass, or k1s/k1e as markers around the f32 kernel's launch(es).
  hipEvent_t k1s = nullptr, k1e = nullptr, k2s = nullptr, k2e = nullptr, t_end = nullptr;
  bool scatter_pending = false;
};

struct Pipe {
  int device = 0;
  int nb = 0, r = 0, dim = 0, bb = 0;
  int tie_cap = 0, flag_cap = 0;
  int64_t pairs_cap = 0;
  hipStream_t side = nullptr, aux = nullptr;
  double tie_rate = 0.004;         // ties per row seen lately (sizes the speculative copy; measured 0.0026 at tau_ulps = 8)
  hipEvent_t done = nullptr;
  int32_t* d_counts = nullptr;     // device int32[2 * counts_cap]: (tie count, stage-1 count) per chunk of a call
  int counts_cap = 0;
  int zeroed_chunks = 0;           // leading counter pairs already zeroed (behind `zeroed`) by the end of the last call
  hipEvent_t zeroed = nullptr;
  int export_blocks = kExportBlocks;
  int export_main = 2;             // trailing chunks whose export stays on the caller's stream
  Slot slot[kSlots];
};

inline int64_t now_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch())
      .count();
}

void pipe_free(Pipe* p) {
  if (p == nullptr) return;
  for (Slot& s : p->slot) {
    if (s.tie_list) (void)hipFree(s.tie_list);
    if (s.flag_list) (void)hipFree(s.flag_list);
    if (s.d_head) (void)hipFree(s.d_head);
    if (s.d_rows) (void)hipFree(s.d_rows);
    if (s.h_head) (void)hipHostFree(s.h_head);
    if (s.h_rows) (void)hipHostFree(s.h_rows);
    if (s.h_patch) (void)hipHostFree(s.h_patch);
    for (hipEvent_t e : {s.exported, s.scattered, s.k1s, s.k1e, s.k2s, s.k2e, s.t_end})
      if (e) (void)hipEventDestroy(e);
  }
  if (p->d_counts) (void)hipFree(p->d_counts);
  if (p->done) (void)hipEventDestroy(p->done);
  if (p->zeroed) (void)hipEventDestroy(p->zeroed);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->aux) (void)hipStreamDestroy(p->aux);
  delete p;
}

#define PIPE_TRY(expr)                    \
  do {                                    \
    const hipError_t e_ = (expr);         \
    if (e_ != hipSuccess) {               \
      rc = -(int)e_;                      \
      goto fail;                          \
    }                                     \
  } while (0)

}  // namespace

extern "C" {

void* lshrs_pipe_create(int32_t num_bands, int32_t rows_per_band, int32_t dim, int32_t tie_cap, int32_t flag_cap) {
  if (num_bands <= 0 || rows_per_band <= 0 || dim <= 0 || tie_cap <= 0 || flag_cap < 0) return nullptr;
  Pipe* p = new (std::nothrow) Pipe();
  if (p == nullptr) return nullptr;
  int rc = 0;
  p->nb = num_bands;
  p->r = rows_per_band;
  p->dim = dim;
  p->bb = (rows_per_band + 7) / 8;
  p->tie_cap = tie_cap;
  p->flag_cap = flag_cap;
  // a 32-column word of the tie list touches up to 32 / (columns per band) bands
  {
    const int per_word = 32 / (8 * p->bb) > 1 ? 32 / (8 * p->bb) : 1;
    p->pairs_cap = ((int64_t)tie_cap * per_word + 7) / 8 * 8;
  }
  if (const char* v = getenv("LSHRS_EXPORT_BLOCKS")) {   // A/B knob
    const int b = atoi(v);
    if (b >= 1 && b <= 4096) p->export_blocks = b;
  }
  if (const char* v = getenv("LSHRS_EXPORT_MAIN")) {     // A/B knob
    const int b = atoi(v);
    if (b >= 1) p->export_main = b;
  }
  PIPE_TRY(hipGetDevice(&p->device));
  {
    int lo = 0, hi = 0;   // (numerically lowest = highest priority)
    PIPE_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* v = getenv("LSHRS_SIDE_PRIORITY");          // A/B knob: "low" / "normal"; default high
    const int prio = (v != nullptr && v[0] == 'l') ? lo : ((v != nullptr && v[0] == 'n') ? 0 : hi);
    PIPE_TRY(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prio));
  }
  PIPE_TRY(hipStreamCreateWithFlags(&p->aux, hipStreamNonBlocking));
  PIPE_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
  PIPE_TRY(hipEventCreateWithFlags(&p->zeroed, hipEventDisableTiming));
  for (Slot& s : p->slot) {
    PIPE_TRY(hipMalloc(&s.tie_list, sizeof(int64_t) * 2 * (size_t)tie_cap));
    if (flag_cap > 0) PIPE_TRY(hipMalloc(&s.flag_list, sizeof(int64_t) * (size_t)flag_cap));
    PIPE_TRY(hipMalloc(&s.d_head, 64 + sizeof(int64_t) * 2 * (size_t)tie_cap));
    PIPE_TRY(hipMalloc(&s.d_rows, sizeof(float) * (size_t)tie_cap * dim));
    PIPE_TRY(hipHostMalloc(&s.h_head, 64 + sizeof(int64_t) * 2 * (size_t)tie_cap, hipHostMallocDefault));
    s.h_counts = reinterpret_cast<int32_t*>(s.h_head);
    s.h_entries = reinterpret_cast<int64_t*>(s.h_head + 64);
    PIPE_TRY(hipHostMalloc(&s.h_rows, sizeof(float) * (size_t)tie_cap * dim, hipHostMallocDefault));
    PIPE_TRY(hipHostMalloc(&s.h_patch, (size_t)(12 + p->bb) * p->pairs_cap, hipHostMallocDefault));
    PIPE_TRY(hipEventCreateWithFlags(&s.exported, hipEventDisableTiming));
    PIPE_TRY(hipEventCreateWithFlags(&s.scattered, hipEventDisableTiming));
    PIPE_TRY(hipEventCreate(&s.k1s));
    PIPE_TRY(hipEventCreate(&s.k1e));
    PIPE_TRY(hipEventCreate(&s.k2s));
    PIPE_TRY(hipEventCreate(&s.k2e));
    PIPE_TRY(hipEventCreate(&s.t_end));
  }
  return p;
fail:
  (void)rc;
  pipe_free(p);
  return nullptr;
}

void lshrs_pipe_destroy(void* pipe) {
  Pipe* p = static_cast<Pipe*>(pipe);
  if (p == nullptr) return;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->side);
  pipe_free(p);
  if (have) (void)hipSetDevice(cur);
}

int lshrs_pipe_hash_f32(void* pipe, const float* X, int64_t ldx, const void* workspace, uint8_t* keys,
                        uint8_t* row_flags, float tau, float tau1, const int64_t* bounds, const uint8_t* chunk_split,
                        int32_t n_chunks, lshrs_tie_resolve_fn resolve, void* engine, const float* planes,
                        int32_t* chunk_status, float* chunk_ms, int64_t* stats, void* stream) {
  const int64_t t_entry = now_ns();
  Pipe* p = static_cast<Pipe*>(pipe);
  if (p == nullptr || X == nullptr || workspace == nullptr || keys == nullptr || bounds == nullptr ||
      chunk_split == nullptr || resolve == nullptr || planes == nullptr || chunk_status == nullptr || n_chunks < 0 ||
      ldx < p->dim)
    return LSHRS_E_BADARG;
  if (stats != nullptr)
    for (int i = 0; i < LSHRS_PIPE_STATS; ++i) stats[i] = 0;
  if (n_chunks == 0) return 0;
  if (bounds[0] != 0) return LSHRS_E_BADARG;
  for (int c = 0; c < n_chunks; ++c) {
    if (bounds[c + 1] <= bounds[c]) return LSHRS_E_BADARG;
    if (chunk_split[c] != 0 && p->flag_cap <= 0) return LSHRS_E_BADARG;
    chunk_status[c] = 0;
  }
  hipStream_t main = static_cast<hipStream_t>(stream);
  const int row_bytes = p->nb * p->bb;
  const bool vec = (p->dim % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  // (the split entry point hands anything else to the f32 kernel itself; deciding it here keeps the timing honest)
  const bool split_ok = vec && (p->dim % 32 == 0) && ldx < (1 << 20);
  int rc = 0;
  int64_t s_ties = 0, s_pairs = 0, s_flagmax = 0, t_head = 0, t_enq = 0, t_wait = 0, t_res = 0, t_scat = 0, t_tail = 0, t_res_last = 0, s_topups = 0;

  {
    int cur = 0;
    PIPE_TRY(hipGetDevice(&cur));
    if (cur != p->device) return LSHRS_E_BADARG;
  }
  if (n_chunks > p->counts_cap) {
    // (grows only on a batch with more chunks than any before it; nothing of an earlier call is in flight on it:
    // every export of that call was waited for)
    if (p->d_counts) (void)hipFree(p->d_counts);
    p->d_counts = nullptr;
    p->counts_cap = 0;
    p->zeroed_chunks = 0;
    const int want = n_chunks < 64 ? 64 : n_chunks;
    PIPE_TRY(hipMalloc(&p->d_counts, sizeof(int32_t) * 2 * (size_t)want));
    p->counts_cap = want;
  }
  // The counters of a call are zeroed at the END of the call before, on the side stream (two fill kernels and their
  // dispatch gaps are ~15 us in front of the first signature pass otherwise); only a first or larger batch fills here.
  if (n_chunks <= p->zeroed_chunks)
    PIPE_TRY(hipStreamWaitEvent(main, p->zeroed, 0));
  else
    PIPE_TRY(hipMemsetAsync(p->d_counts, 0, sizeof(int32_t) * 2 * (size_t)n_chunks, main));
  p->zeroed_chunks = 0;

  {
    auto enqueue = [&](int c) -> int {
      Slot& s = p->slot[c % kSlots];
      const int64_t lo = bounds[c], hi = bounds[c + 1];
      const float* xs = X + lo * ldx;
      uint8_t* ks = keys + lo * row_bytes;
      uint8_t* fl = row_flags != nullptr ? row_flags + lo : nullptr;
      int32_t* cnt = p->d_counts + 2 * (size_t)c;
      hipError_t e;
      // Times, when asked for.  Split pass: start/stop events that ride on the dispatch packets of its two kernels
      // (what a kernel trace reports; no packet of their own in the stream).  f32 kernel (one or two launches): a
      // marker event on either side of the call.
      const bool split = chunk_split[c] != 0 && split_ok;
      int r;
      if (split) {
        lshrs_sig_opts o{};
        o.struct_bytes = sizeof(o);
        o.ev_stage1_start = s.k1s;
        o.ev_stage1_stop = s.k1e;
        o.ev_stage2_start = s.k2s;
        o.ev_stage2_stop = s.k2e;
        r = lshrs_sig_hash_batch_split_f32(xs, hi - lo, ldx, workspace, p->nb, p->r, p->dim, ks, s.tie_list, p->tie_cap,
                                           cnt, tau, fl, s.flag_list, p->flag_cap, cnt + 1, tau1,
                                           chunk_ms != nullptr ? &o : nullptr, stream);
      } else {
        if (chunk_ms != nullptr && (e = hipEventRecord(s.k1s, main)) != hipSuccess) return -(int)e;
        r = lshrs_sig_hash_batch_f32(xs, hi - lo, ldx, workspace, p->nb, p->r, p->dim, ks, s.tie_list, p->tie_cap, cnt,
                                     tau, fl, nullptr, stream);
        if (chunk_ms != nullptr && (e = hipEventRecord(s.k1e, main)) != hipSuccess) return -(int)e;
      }
      if (r != 0) return r;
      // (Tried: stage 2 of the mid-batch chunks on the side stream, behind the stop event that rides on stage 1's
      // dispatch, so that the stage-1 kernels follow each other directly.  Correct, and slower - 1.42-1.48 against
      // 1.39-1.44 ms per step: the fix-up's latency-bound waves, 48 KiB of LDS each, keep whole CUs from the pass
      // beside them for ~100 us per chunk, far more than the two dispatch gaps they were meant to save.)
      if ((e = hipEventRecord(s.t_end, main)) != hipSuccess) return -(int)e;
      // The export runs beside the next chunk's signature pass on the side stream — where it gets no CU before that
      // pass's first round of workgroups retires (~70 us; they take whole register files).  Harmless in mid-batch,
      // but the export of the last-but-one chunk would arrive when the last kernel is ending and leave the host TWO
      // chunks to resolve behind the GPU: the last two exports stay on the caller's stream (~11 us of GPU time for
      // the one, nothing follows the other), which also skips the cross-stream hand-over (~15 us).
      hipStream_t xstream = (c >= n_chunks - p->export_main) ? main : p->side;
      const bool on_main = xstream == main;
      if (xstream != main && (e = hipStreamWaitEvent(p->side, s.t_end, 0)) != hipSuccess) return -(int)e;
      // The exports of the last two chunks, on the caller's stream, write pinned host memory themselves (nothing runs
      // beside them, and it is the shortest way to the host).  Every other export lands in device memory and the
      // runtime's copy carries a SPECULATIVE prefix of it to the host - sized from the tie rate of the chunks before,
      // topped up by the host in the rare case it falls short: a kernel that stores to host memory while the
      // signature pass runs slows that pass down (measured: 70 us per 1M rows; the runtime's copies much less).
      const bool direct = on_main;
      int64_t* o_entries = direct ? s.h_entries : reinterpret_cast<int64_t*>(s.d_head + 64);
      float* o_rows = direct ? s.h_rows : s.d_rows;
      int32_t* o_counts = direct ? s.h_counts : reinterpret_cast<int32_t*>(s.d_head);
      const int blocks = p->export_blocks;
      if (vec)
        hipLaunchKernelGGL(export_ties_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, xstream, xs, ldx, p->dim,
                           s.tie_list, cnt, p->tie_cap, o_entries, o_rows, o_counts);
      else
        hipLaunchKernelGGL(export_ties_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, xstream, xs, ldx, p->dim,
                           s.tie_list, cnt, p->tie_cap, o_entries, o_rows, o_counts);
      if ((e = hipGetLastError()) != hipSuccess) return -(int)e;
      if (direct) {
        s.copied = p->tie_cap;
      } else {
        int64_t pred = (int64_t)((double)(hi - lo) * p->tie_rate * 1.5) + 128;
        if (pred > p->tie_cap) pred = p->tie_cap;
        s.copied = pred;
        if ((e = hipMemcpyAsync(s.h_head, s.d_head, 64 + 16 * (size_t)pred, hipMemcpyDeviceToHost, p->side)) != hipSuccess)
          return -(int)e;
        if ((e = hipMemcpyAsync(s.h_rows, s.d_rows, sizeof(float) * (size_t)pred * p->dim, hipMemcpyDeviceToHost,
                                p->side)) != hipSuccess)
          return -(int)e;
      }
      if ((e = hipEventRecord(s.exported, xstream)) != hipSuccess) return -(int)e;
      return 0;
    };

    for (int c = 0; c < n_chunks && c < kAhead; ++c) {
      if ((rc = enqueue(c)) != 0) goto fail;
      if (c == 0) t_head = now_ns() - t_entry;
    }
    for (int c = 0; c < n_chunks; ++c) {
      Slot& s = p->slot[c % kSlots];
      int64_t t0 = now_ns();
      PIPE_TRY(hipEventSynchronize(s.exported));
      int64_t t1 = now_ns();
      t_wait += t1 - t0;
      if (c == n_chunks - 1) t_tail = t1 - t_entry;
      if (chunk_ms != nullptr) {
        float a = -1.f, b = -1.f;     // (this slot is re-armed kSlots chunks on: read now)
        // Diagnostics must not fail the batch.
        if (hipEventElapsedTime(&a, s.k1s, s.k1e) != hipSuccess) a = -1.f;
        if (chunk_split[c] != 0 && split_ok && hipEventElapsedTime(&b, s.k2s, s.k2e) != hipSuccess) b = -1.f;
        (void)hipGetLastError();
        chunk_ms[2 * c] = a;
        chunk_ms[2 * c + 1] = b;
      }
      if (c + kAhead < n_chunks) {
        if ((rc = enqueue(c + kAhead)) != 0) goto fail;
      }
      t0 = now_ns();
      t_enq += t0 - t1;
      const int64_t wanted = s.h_counts[0], flagged = s.h_counts[1];
      if (flagged > s_flagmax) s_flagmax = flagged;
      if (chunk_split[c] && flagged > p->flag_cap) {
        chunk_status[c] = 2;
        continue;
      }
      if (wanted > p->tie_cap) {
        chunk_status[c] = 1;
        continue;
      }
      if (wanted <= 0) continue;
      p->tie_rate = 0.75 * p->tie_rate + 0.25 * (double)wanted / (double)(bounds[c + 1] - bounds[c]);
      if (wanted > s.copied) {   // the speculative copy fell short: fetch the rest (the export itself is complete)
        const int64_t more = wanted - s.copied;
        PIPE_TRY(hipMemcpyAsync(s.h_entries + 2 * s.copied, s.d_head + 64 + 16 * s.copied, 16 * (size_t)more,
                                hipMemcpyDeviceToHost, p->aux));
        PIPE_TRY(hipMemcpyAsync(s.h_rows + s.copied * p->dim, s.d_rows + s.copied * p->dim,
                                sizeof(float) * (size_t)more * p->dim, hipMemcpyDeviceToHost, p->aux));
        PIPE_TRY(hipStreamSynchronize(p->aux));
        ++s_topups;
      }
      if (s.scatter_pending) {   // (kSlots chunks ago: long finished)
        PIPE_TRY(hipEventSynchronize(s.scattered));
        s.scatter_pending = false;
      }
      int64_t* p_rows = reinterpret_cast<int64_t*>(s.h_patch);
      int32_t* p_bands = reinterpret_cast<int32_t*>(s.h_patch + 8 * p->pairs_cap);
      uint8_t* p_keys = s.h_patch + 12 * p->pairs_cap;
      int64_t m = 0;
      const int rr = resolve(engine, planes, p->nb, p->r, p->dim, s.h_entries, wanted, s.h_rows, p->dim, p_rows, p_bands,
                             p_keys, p->pairs_cap, &m);
      t1 = now_ns();
      t_res += t1 - t0;
      if (c == n_chunks - 1) t_res_last = t1 - t0;
      if (rr != 0) {             // (more pairs than the patch slot holds: the caller redoes the chunk with room)
        chunk_status[c] = 1;
        continue;
      }
      s_ties += wanted;
      s_pairs += m;
      if (m > 0) {
        if ((rc = lshrs_scatter_band_keys_u8(keys + bounds[c] * row_bytes, p->nb, p->bb, p_rows, p_bands, p_keys, m,
                                             p->side)) != 0)
          goto fail;
        PIPE_TRY(hipEventRecord(s.scattered, p->side));
        s.scatter_pending = true;
      }
      t_scat += now_ns() - t1;
    }
    PIPE_TRY(hipEventRecord(p->done, p->side));
    PIPE_TRY(hipStreamWaitEvent(main, p->done, 0));
    // every counter of this call has been exported: zero them for the next call, off the caller's stream
    PIPE_TRY(hipMemsetAsync(p->d_counts, 0, sizeof(int32_t) * 2 * (size_t)n_chunks, p->side));
    PIPE_TRY(hipEventRecord(p->zeroed, p->side));
    p->zeroed_chunks = n_chunks;
  }
  if (stats != nullptr) {
    stats[0] = s_ties;
    stats[1] = s_pairs;
    stats[2] = s_flagmax;
    stats[3] = t_head;
    stats[4] = t_enq;
    stats[5] = t_wait;
    stats[6] = t_res;
    stats[7] = t_scat;
    stats[8] = t_tail;
    stats[9] = now_ns() - t_entry;
    stats[10] = t_res_last;
    stats[11] = s_topups;
  }
  return 0;

fail:
  // kernels and exports still in flight use this object's scratch and the caller's buffers: let them finish
  (void)hipStreamSynchronize(main);
  (void)hipStreamSynchronize(p->side);
  for (Slot& s : p->slot) s.scatter_pending = false;
  p->zeroed_chunks = 0;
  (void)hipGetLastError();
  return rc;
}

}  // extern "C"
