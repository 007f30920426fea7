// Host tie-break engine: the reference's own BLAS call, on several cores at once.
//
// The signature kernel reports every projection whose magnitude is too small for its sign to be trusted
// against the reference's summation order.  Those (row, band) pairs are re-evaluated with
// the expression the reference itself uses, `projection @ vector` (lshrs/hash/lsh.py:200), i.e. one
// cblas_sgemv(RowMajor, NoTrans, rows_per_band, dim) of NumPy's bundled OpenBLAS per pair.  That costs ~0.9 us
// per pair on one core and is what bounds the bit-exact ingest rate.
//
// OpenBLAS serialises concurrent callers on a process-wide buffer lock, so threads calling into ONE copy of the
// library do not scale.  This engine therefore maps the library file N times (memfd copies: distinct inodes, so
// the dynamic loader gives each copy its own globals and its own lock) and gives each worker thread a private
// copy.  Same machine code, same kernel selection, same operands -> the same bits as NumPy's call
// (tests/test_tiebreak_host.py compares them, and lshrs_amd/hasher.py self-checks each shape before trusting it).
//
// Plain C ABI (include/lshrs_host.h); no HIP, no torch.  Built with g++ by lshrs_amd/_native.py.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lshrs_host.h"

namespace {

// cblas_sgemv of an ILP64 OpenBLAS (NumPy >= 2 bundles libscipy_openblas64_: 64-bit integers)
using sgemv64_fn = void (*)(int order, int trans, int64_t m, int64_t n, float alpha, const float* a, int64_t lda,
                            const float* x, int64_t incx, float beta, float* y, int64_t incy);
// ... and of an LP64 build (32-bit integers)
using sgemv32_fn = void (*)(int order, int trans, int m, int n, float alpha, const float* a, int lda, const float* x,
                            int incx, float beta, float* y, int incy);
using set_threads_fn = void (*)(int);

constexpr int kRowMajor = 101, kNoTrans = 111;

constexpr int64_t kGrain = 16;  // pairs claimed per atomic fetch (large jobs; a small job is cut finer, Job::grain)

struct Job {
  const float* planes = nullptr;
  int r = 0, dim = 0, band_bytes = 0;
  const float* xrows = nullptr;
  int64_t ldx = 0;
  const int32_t* row_index = nullptr;
  const int32_t* band = nullptr;
  int64_t n_pairs = 0;
  uint8_t* out_keys = nullptr;
  float* out_y = nullptr;  // optional: the projections themselves (tests)
  int64_t grain = kGrain;
};

struct PairRec {
  int64_t row;
  int32_t band;
  int32_t entry;
};

struct Engine {
  int ilp64 = 1;
  std::vector<void*> handles;
  std::vector<void*> sgemv;
  std::vector<int> fds;
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t generation = 0;
  std::atomic<uint64_t> gen_atomic{0};     // mirror of generation for the lock-free poll
  std::atomic<bool> stop_atomic{false};
  std::atomic<int> pending{0};             // workers still inside the current job
  bool stop = false;
  Job job;
  std::atomic<int64_t> next{0};
  std::mutex run_mu;  // one lshrs_tb_patch / lshrs_tb_resolve at a time
  std::vector<PairRec> scratch_pairs;
  std::vector<int32_t> scratch_index;
};


void run_pairs(Engine* e, int t) {
  const Job& j = e->job;
  float ybuf[64];
  std::vector<float> ybig;
  float* y = ybuf;
  if (j.r > 64) {
    ybig.resize(j.r);
    y = ybig.data();
  }
  for (;;) {
    const int64_t lo = e->next.fetch_add(j.grain, std::memory_order_relaxed);
    if (lo >= j.n_pairs) break;
    const int64_t hi = lo + j.grain < j.n_pairs ? lo + j.grain : j.n_pairs;
    for (int64_t p = lo; p < hi; ++p) {
      const float* plane = j.planes + (size_t)j.band[p] * j.r * j.dim;
      const float* x = j.xrows + (size_t)j.row_index[p] * j.ldx;
      if (e->ilp64)
        reinterpret_cast<sgemv64_fn>(e->sgemv[t])(kRowMajor, kNoTrans, j.r, j.dim, 1.0f, plane, j.dim, x, 1, 0.0f, y, 1);
      else
        reinterpret_cast<sgemv32_fn>(e->sgemv[t])(kRowMajor, kNoTrans, j.r, j.dim, 1.0f, plane, j.dim, x, 1, 0.0f, y, 1);
      uint8_t* out = j.out_keys + (size_t)p * j.band_bytes;
      for (int b = 0; b < j.band_bytes; ++b) {  // np.packbits(y > 0, bitorder="little"): lsh.py:204-208
        unsigned v = 0;
        for (int k = 0; k < 8 && 8 * b + k < j.r; ++k) v |= (y[8 * b + k] > 0.0f ? 1u : 0u) << k;
        out[b] = (uint8_t)v;
      }
      if (j.out_y != nullptr) memcpy(j.out_y + (size_t)p * j.r, y, sizeof(float) * j.r);
    }
  }
}

void worker_main(Engine* e, int t) {
  uint64_t seen = 0;
  for (;;) {
    // Chunks of one batch arrive 100-400 microseconds apart: poll for a millisecond before going to sleep, a
    // condition-variable wake-up alone costs as much as the work of a small chunk.
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(1000);
    while (e->gen_atomic.load(std::memory_order_acquire) == seen && !e->stop_atomic.load(std::memory_order_relaxed) &&
           std::chrono::steady_clock::now() < spin_until)
      __builtin_ia32_pause();
    {
      std::unique_lock<std::mutex> lk(e->mu);
      e->cv_go.wait(lk, [&] { return e->stop || e->generation != seen; });
      if (e->stop) return;
      seen = e->generation;
    }
    run_pairs(e, t);
    if (e->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
      std::lock_guard<std::mutex> lk(e->mu);   // (the caller may have stopped polling and gone to sleep on cv_done)
      e->cv_done.notify_all();
    }
  }
}

void* map_private_copy(const char* path, int* fd_out) {
  const int in = open(path, O_RDONLY | O_CLOEXEC);
  if (in < 0) return nullptr;
  const int fd = memfd_create("lshrs_blas_copy", MFD_CLOEXEC);
  if (fd < 0) {
    close(in);
    return nullptr;
  }
  std::vector<char> buf(1 << 20);
  for (;;) {
    const ssize_t got = read(in, buf.data(), buf.size());
    if (got < 0) {
      close(in);
      close(fd);
      return nullptr;
    }
    if (got == 0) break;
    ssize_t off = 0;
    while (off < got) {
      const ssize_t put = write(fd, buf.data() + off, (size_t)(got - off));
      if (put <= 0) {
        close(in);
        close(fd);
        return nullptr;
      }
      off += put;
    }
  }
  close(in);
  const std::string proc = "/proc/self/fd/" + std::to_string(fd);
  // RTLD_DEEPBIND: the copy binds its internal references to ITS OWN globals, not to the first copy loaded.
  // Its dependencies (libgfortran, libquadmath) are found by SONAME among the libraries NumPy already loaded.
  void* h = dlopen(proc.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
  if (h == nullptr) {
    close(fd);
    return nullptr;
  }
  *fd_out = fd;
  return h;
}

}  // namespace

extern "C" {

int lshrs_host_abi_version(void) { return LSHRS_HOST_ABI_VERSION; }

void* lshrs_tb_create(const char* blas_path, const char* sgemv_symbol, const char* set_threads_symbol, int ilp64,
                      int n_threads) {
  if (blas_path == nullptr || sgemv_symbol == nullptr || n_threads < 1 || n_threads > 64) return nullptr;
  Engine* e = new Engine();
  e->ilp64 = ilp64 ? 1 : 0;
  // OpenBLAS starts a pool of (cores - 1) busy-waiting threads when it is LOADED.  Eight copies on a 256-core
  // host are ~500 spinning threads for the first tenth of a second - enough to exhaust a container's CPU quota
  // and get the whole process throttled (measured: 300-480 ms stalls in the first hashing steps).  The copies are
  // only ever called from one thread each, so they are loaded with the thread count pinned to 1: no pools at all.
  static const char* const kThreadVars[] = {"OPENBLAS_NUM_THREADS", "GOTO_NUM_THREADS", "OMP_NUM_THREADS"};
  std::string saved[3];
  bool had[3];
  for (int v = 0; v < 3; ++v) {
    const char* cur = getenv(kThreadVars[v]);
    had[v] = cur != nullptr;
    if (had[v]) saved[v] = cur;
    setenv(kThreadVars[v], "1", 1);
  }
  for (int t = 0; t < n_threads; ++t) {
    int fd = -1;
    void* h = map_private_copy(blas_path, &fd);
    void* fn = h ? dlsym(h, sgemv_symbol) : nullptr;
    if (fn == nullptr) {
      if (h) dlclose(h);
      if (fd >= 0) close(fd);
      break;
    }
    if (set_threads_symbol != nullptr && set_threads_symbol[0] != '\0') {
      void* st = dlsym(h, set_threads_symbol);
      if (st != nullptr) reinterpret_cast<set_threads_fn>(st)(1);  // each copy serves exactly one caller
    }
    e->handles.push_back(h);
    e->sgemv.push_back(fn);
    e->fds.push_back(fd);
  }
  for (int v = 0; v < 3; ++v) {
    if (had[v]) setenv(kThreadVars[v], saved[v].c_str(), 1);
    else unsetenv(kThreadVars[v]);
  }
  if (e->handles.empty()) {
    delete e;
    return nullptr;
  }
  const int got = (int)e->handles.size();
  for (int t = 1; t < got; ++t) e->workers.emplace_back(worker_main, e, t);  // the caller is worker 0
  return e;
}

// Model of the host BLAS's summation order for one dot product (the order sig_fix8_kernel<true> replays on the GPU).
// model 1 = OpenBLAS's sgemv_t on x86-64 (Haswell / Zen / SkylakeX builds), as `P_band @ x` of an r-row band reaches it:
//   * rows are taken four at a time by the 8-lane AVX kernel ("kind 0"): eight interleaved single-rounded fma chains
//     p_j = sum over k = j (mod 8) of a_k x_k, reduced as ((p0+p4) + (p1+p5)) + ((p2+p6) + (p3+p7));
//   * the r mod 4 rows that are left go through the SSE kernels, which multiply and add in two roundings: a pair of rows
//     through the 4x2 kernel ("kind 1": four chains over k = l (mod 4), reduced (v0+v1) + (v2+v3)), a single row -
//     the only one left, or the third of three - through the 4x1 kernel ("kind 2": the eight chains and the tree of
//     kind 0, unfused);
//   * the vector is consumed in blocks of 4096 elements: every block's sum is reduced on its own and added to y;
//   * a vector of 8 m + 4 elements (300-d, 100-d): the 8-lane kernels take the first four with their low lanes and then
//     eight at a time from the fifth (modelled for n <= 4096).
//   * the n mod 4 elements behind the last whole group of four are added in plain C after all blocks (sgemv_t.c's "m3"
//     branches as gcc contracts them): one left, y = fma(a0, x0, y); two, y = y + fma(a0, x0, fl(a1 x1)); three,
//     y = y + fma(a2, x2, fma(a0, x0, fl(a1 x1))) - found by search on this library (tools/blas_order/), n >= 9: shorter
//     vectors do not reach these kernels the same way and are refused.
//     model 2 = the same with the tail as the library's Haswell / Zen build compiles it: y + fl(a0 x0), y + (fl(a0 x0) +
//     fl(a1 x1)), y + ((fl(a0 x0) + fl(a1 x1)) + fl(a2 x2)) - nothing contracted.  (n % 4 == 0: models 1 and 2 coincide.)
// rows_per_band = 1 is another kernel (sdot: tb_model_sdot below).  Found by search (tools/blas_order/); the caller (lshrs_amd/_hostblas.py) compares it bit for
// bit with `P_band @ x` of the running process, row kind by row kind, before the device replay may stand in for the host.
static inline int tb_row_kind(int row, int rows) {
  const int r4 = rows & ~3;
  if (row < r4) return 0;
  return ((rows & 3) == 1 || row - r4 == 2) ? 2 : 1;
}

// A band of ONE row: NumPy's matmul sends (1, n) @ (n,) to sdot, not to sgemv.  OpenBLAS's x86-64 sdot hands the first
// n1 = n & -32 elements to a SIMD kernel with an f32 result, sums the f32 PRODUCTS of the n - n1 elements behind them one by one
// in a DOUBLE, adds the kernel's result to that double and rounds once to f32 (found in round 5 by masking probes, FPRev style,
// tools/blas_order/fprev.py; checked bit for bit for every n from 1 to 139 and beyond under OPENBLAS_CORETYPE=SkylakeX / Haswell).
// The two builds of the library differ in the kernel:
//   model 1 (SkylakeX build): whole 64-element steps on four 16-lane fma accumulators, each folded to eight lanes
//     (l + (l + 8)); a last 32-element step (n1 % 64 == 32) goes straight onto the four folded accumulators, eight lanes
//     each; the four added in turn ((t0 + t1) + t2) + t3, lanes i + (i + 4), then (w0 + w1) + (w2 + w3);
//   model 2 (Haswell / Zen build): 32-element steps on four 8-lane fma accumulators, each folded to four lanes
//     (l + (l + 4)), added pairwise (q0 + q1) + (q2 + q3), then (l0 + l1) + (l2 + l3).
static float tb_model_sdot(const float* a, const float* x, int64_t n, int32_t model) {
  const int64_t n1 = n & ~(int64_t)31;
  float kernel = 0.f;
  if (n1 > 0 && model == 1) {
    const int64_t n64 = n1 & ~(int64_t)63;
    float acc[64] = {0};
    for (int64_t k = 0; k < n64; ++k) acc[k & 63] = __builtin_fmaf(a[k], x[k], acc[k & 63]);
    float t[4][8];
    for (int u = 0; u < 4; ++u)
      for (int l = 0; l < 8; ++l) t[u][l] = n64 > 0 ? acc[16 * u + l] + acc[16 * u + l + 8] : 0.f;
    if (n64 < n1)
      for (int u = 0; u < 4; ++u)
        for (int l = 0; l < 8; ++l) t[u][l] = __builtin_fmaf(a[n64 + 8 * u + l], x[n64 + 8 * u + l], t[u][l]);
    float v[8];
    for (int l = 0; l < 8; ++l) v[l] = ((t[0][l] + t[1][l]) + t[2][l]) + t[3][l];
    const float w0 = v[0] + v[4], w1 = v[1] + v[5], w2 = v[2] + v[6], w3 = v[3] + v[7];
    kernel = (w0 + w1) + (w2 + w3);
  } else if (n1 > 0) {
    float acc[32] = {0};
    for (int64_t k = 0; k < n1; ++k) acc[k & 31] = __builtin_fmaf(a[k], x[k], acc[k & 31]);
    float v[4];
    for (int l = 0; l < 4; ++l) {
      const float s0 = acc[l] + acc[4 + l], s1 = acc[8 + l] + acc[12 + l], s2 = acc[16 + l] + acc[20 + l], s3 = acc[24 + l] + acc[28 + l];
      v[l] = (s0 + s1) + (s2 + s3);
    }
    kernel = (v[0] + v[1]) + (v[2] + v[3]);
  }
  double tail = 0.0;
  for (int64_t k = n1; k < n; ++k) {
    volatile float prod = a[k] * x[k];      // (an f32 product, rounded on its own: nothing here may be contracted)
    tail += (double)prod;
  }
  return (float)(tail + (double)kernel);
}

// model 3: OpenBLAS's SkylakeX build on a contiguous band of two rows and more over at most EIGHT elements - its small-matrix
// kernels (sgemv_t_microk_skylakex: lda == m <= 8).  The band's rows go in blocks of 16 / 8, then one of four, a pair, a single
// row - each with arithmetic of its own per length.  Found with tools/blas_order/small_matrix_search.py (summation tree by
// masking probes, then every placement of fused multiply-adds on it against NumPy bit for bit; profiles/r05_blas_sdot_order.log
// part 5), checked for bands of 2 .. 64 rows in tests/test_reference_blas.py.  p(k) = fl(a_k x_k); f(k, s) = fma(a_k, x_k, s).
static float tb_model_small_skx(const float* a, const float* x, int n, int row, int rows) {
  auto p = [&](int k) { const volatile float v = a[k] * x[k]; return (float)v; };
  auto f = [&](int k, float s) { return __builtin_fmaf(a[k], x[k], s); };
  auto add = [](float u, float v) { const volatile float s = u + v; return (float)s; };
  const int b16 = rows / 16 * 16, b8 = rows / 8 * 8;
  int rem = rows - b8;
  int type;                                   // 16, 8, 4, 2, 1: the block this row sits in
  if (row < b16) type = 16;
  else if (row < b8) type = 8;
  else if (rem >= 4 && row < b8 + 4) type = 4;
  else {
    const int base = b8 + (rem >= 4 ? 4 : 0);
    rem -= rem >= 4 ? 4 : 0;
    type = (rem >= 2 && row < base + 2) ? 2 : 1;
  }
  auto seq = [&]() {                          // plain left-to-right adds of rounded products
    float s = p(0);
    for (int k = 1; k < n; ++k) s = add(s, p(k));
    return s;
  };
  auto chain = [&]() {                        // fma chain in element order, started by the rounded first product
    float s = p(0);
    for (int k = 1; k < n; ++k) s = f(k, s);
    return s;
  };
  switch (n) {
    case 1: return p(0);
    case 2: return type == 16 ? f(1, p(0)) : add(p(0), p(1));
    case 3:
      if (type >= 8) return f(2, f(1, p(0)));
      if (type == 2) return add(add(p(0), p(1)), p(2));
      return f(2, f(0, p(1)));                                              // blocks of four and the single row
    case 4: return type == 1 ? seq() : add(add(p(0), p(1)), add(p(2), p(3)));
    case 5: return type >= 4 ? chain() : seq();
    case 6:
      if (type >= 8) return chain();
      if (type == 4) return add(add(p(0), f(1, p(2))), add(p(3), f(4, p(5))));
      return seq();
    case 7:
      if (type >= 8) return chain();
      if (type == 4) return add(add(f(0, p(1)), f(4, p(5))), add(f(2, p(3)), p(6)));
      return seq();
    default:                                                                // 8
      if (type >= 4) return add(add(add(p(0), p(1)), add(p(2), p(3))), add(add(p(4), p(5)), add(p(6), p(7))));
      if (type == 2) return add(add(add(p(0), p(1)), add(p(4), p(5))), add(add(p(2), p(3)), add(p(6), p(7))));
      return add(add(add(add(p(0), p(4)), add(p(1), p(5))), add(p(2), p(6))), add(p(3), p(7)));
  }
}

float lshrs_tb_model_row_dot(const float* a, const float* x, int64_t n, int32_t model, int32_t row, int32_t rows_per_band) {
  if ((model != 1 && model != 2 && model != 3) || a == nullptr || x == nullptr || n <= 0 || row < 0 || row >= rows_per_band)
    return __builtin_nanf("");
  if (model == 3)
    return (n <= 8 && rows_per_band >= 2) ? tb_model_small_skx(a, x, (int)n, row, rows_per_band) : __builtin_nanf("");
  if (rows_per_band == 1) return tb_model_sdot(a, x, n, model);
  // fewer than 9 elements: the SkylakeX build takes small-matrix paths of its own there (seven different trees, not modelled);
  // the Haswell / Zen build (model 2) runs the same kernels as for longer rows - verified for every length from 1 and every row
  // kind against NumPy on that build (tests/test_reference_blas.py)
  if (n % 4 != 0 && (rows_per_band < 2 || (n < 9 && model != 2))) return __builtin_nanf("");
  const int64_t body = n & ~(int64_t)3;                       // whole groups of four: the kernels' share
  // (8 m + 4 elements beyond 4096: the last, short block takes ITS first four first - the loop below does that per block; checked
  //  against NumPy on both builds at 4100, 4108, 5004, 8196, 8204, 12292 elements and with tails: tests/test_reference_blas.py)
  const int kind = tb_row_kind(row, rows_per_band);
  float y = 0.f;
  for (int64_t k0 = 0; k0 < body; k0 += 4096) {
    const int64_t kn = body - k0 < 4096 ? body - k0 : 4096;
    const float* ab = a + k0;
    const float* xb = x + k0;
    float s;
    if (kind == 1) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int64_t k = 0; k < kn; k += 4)
        for (int l = 0; l < 4; ++l) {
          const volatile float prod = ab[k + l] * xb[k + l];      // (volatile: two roundings, whatever the compiler's flags)
          v[l] = v[l] + prod;
        }
      s = (v[0] + v[1]) + (v[2] + v[3]);
    } else {
      // a block of 8 m + 4 elements: the 8-lane kernels take the FIRST four with their low lanes, then eight at a time
      float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int64_t head = kn % 8;
      for (int64_t j = 0; j < head; ++j) {
        const volatile float prod = ab[j] * xb[j];                // (fma(a, x, 0) and 0 + fl(a x) are the same value)
        p[j] = prod;
      }
      if (kind == 0) {
        for (int64_t k = head; k < kn; k += 8)
          for (int j = 0; j < 8; ++j) p[j] = __builtin_fmaf(ab[k + j], xb[k + j], p[j]);
      } else {
        for (int64_t k = head; k < kn; k += 8)
          for (int j = 0; j < 8; ++j) {
            const volatile float prod = ab[k + j] * xb[k + j];
            p[j] = p[j] + prod;
          }
      }
      const float q0 = p[0] + p[4], q1 = p[1] + p[5], q2 = p[2] + p[6], q3 = p[3] + p[7];
      const float h0 = q0 + q1, h1 = q2 + q3;
      s = h0 + h1;
    }
    y = k0 == 0 ? s : y + s;
  }
  const float* at = a + body;
  const float* xt = x + body;
  if (model == 2) {                                           // the Haswell / Zen build of the same file: nothing contracted
    for (int64_t j = 0, m3 = n - body; j < m3 && m3 > 0; ++j) {
      if (m3 == 1) { const volatile float p0 = at[0] * xt[0]; y = y + p0; break; }
      const volatile float p0 = at[0] * xt[0], p1 = at[1] * xt[1];
      volatile float t = p0 + p1;
      if (m3 == 3) { const volatile float p2 = at[2] * xt[2]; t = t + p2; }
      y = y + t;
      break;
    }
    return y;
  }
  switch (n - body) {                                         // the scalar tail (see above)
    case 1: y = __builtin_fmaf(at[0], xt[0], y); break;
    case 2: { const volatile float p1 = at[1] * xt[1]; const volatile float t = __builtin_fmaf(at[0], xt[0], p1); y = y + t; break; }
    case 3: {
      const volatile float p1 = at[1] * xt[1];
      const volatile float t = __builtin_fmaf(at[2], xt[2], __builtin_fmaf(at[0], xt[0], p1));
      y = y + t;
      break;
    }
    default: break;
  }
  return y;
}

// (a row of the 8-lane kernel: what every row of a band of 4, 8, 12 ... rows is)
float lshrs_tb_model_dot(const float* a, const float* x, int64_t n, int32_t model) {
  return lshrs_tb_model_row_dot(a, x, n, model, 0, 4);
}

int lshrs_tb_threads(void* engine) { return engine ? (int)static_cast<Engine*>(engine)->handles.size() : 0; }

void lshrs_tb_destroy(void* engine) {
  if (engine == nullptr) return;
  Engine* e = static_cast<Engine*>(engine);
  {
    std::lock_guard<std::mutex> lk(e->mu);
    e->stop = true;
    e->stop_atomic.store(true);
  }
  e->cv_go.notify_all();
  for (auto& w : e->workers) w.join();
  // The private copies stay mapped: OpenBLAS starts helper threads at load time and unloading a library under
  // them is not safe.  The memfds can go; the mappings keep the pages alive.
  for (int fd : e->fds) close(fd);
  delete e;
}

static int run_job(Engine* e, const float* planes, int32_t rows_per_band, int32_t dim, const float* xrows, int64_t ldx,
                   const int32_t* row_index, const int32_t* band, int64_t n_pairs, uint8_t* out_keys, float* out_y) {
  e->job.planes = planes;
  e->job.r = rows_per_band;
  e->job.dim = dim;
  e->job.band_bytes = (rows_per_band + 7) / 8;
  e->job.xrows = xrows;
  e->job.ldx = ldx;
  e->job.row_index = row_index;
  e->job.band = band;
  e->job.n_pairs = n_pairs;
  e->job.out_keys = out_keys;
  e->job.out_y = out_y;
  // the last chunk of a batch is a small job on the critical path (~170 pairs): four claims per thread keep the
  // threads level where grains of 16 would leave some with twice the work of others
  {
    const int64_t threads = (int64_t)e->workers.size() + 1;
    const int64_t fine = n_pairs / (4 * threads);
    e->job.grain = fine < 1 ? 1 : (fine < kGrain ? fine : kGrain);
  }
  e->next.store(0, std::memory_order_relaxed);
  const bool fan_out = !e->workers.empty() && n_pairs > 4 * kGrain;  // a handful of pairs is cheaper done here
  if (fan_out) {
    {
      std::lock_guard<std::mutex> lk(e->mu);
      e->pending.store((int)e->workers.size(), std::memory_order_relaxed);
      ++e->generation;
      e->gen_atomic.store(e->generation, std::memory_order_release);
    }
    e->cv_go.notify_all();
  }
  run_pairs(e, 0);
  if (fan_out) {
    // the workers finish within microseconds of the caller (they share one work queue): poll before sleeping —
    // a futex wake-up costs as much as a small chunk's whole job
    const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
    while (e->pending.load(std::memory_order_acquire) != 0 && std::chrono::steady_clock::now() < spin_until)
      __builtin_ia32_pause();
    if (e->pending.load(std::memory_order_acquire) != 0) {
      std::unique_lock<std::mutex> lk(e->mu);
      e->cv_done.wait(lk, [&] { return e->pending.load(std::memory_order_acquire) == 0; });
    }
  }
  return 0;
}

int lshrs_tb_patch(void* engine, const float* planes, int32_t num_bands, int32_t rows_per_band, int32_t dim,
                   const float* xrows, int64_t ldx, const int32_t* row_index, const int32_t* band, int64_t n_pairs,
                   uint8_t* out_keys, float* out_y) {
  if (n_pairs == 0) return 0;
  if (engine == nullptr || planes == nullptr || xrows == nullptr || row_index == nullptr || band == nullptr ||
      out_keys == nullptr || n_pairs < 0 || num_bands < 1 || rows_per_band < 1 || dim < 1 || ldx < dim)
    return LSHRS_HOST_E_BADARG;
  for (int64_t p = 0; p < n_pairs; ++p)
    if (band[p] < 0 || band[p] >= num_bands || row_index[p] < 0) return LSHRS_HOST_E_BADARG;
  Engine* e = static_cast<Engine*>(engine);
  std::lock_guard<std::mutex> run(e->run_mu);
  return run_job(e, planes, rows_per_band, dim, xrows, ldx, row_index, band, n_pairs, out_keys, out_y);
}

int lshrs_tb_resolve(void* engine, const float* planes, int32_t num_bands, int32_t rows_per_band, int32_t dim,
                     const int64_t* entries, int64_t n_entries, const float* xstage, int64_t ldx, int64_t* out_rows,
                     int32_t* out_bands, uint8_t* out_keys, int64_t out_cap, int64_t* n_pairs) {
  if (n_pairs == nullptr) return LSHRS_HOST_E_BADARG;
  *n_pairs = 0;
  if (n_entries == 0) return 0;
  if (engine == nullptr || planes == nullptr || entries == nullptr || xstage == nullptr || out_rows == nullptr ||
      out_bands == nullptr || out_keys == nullptr || n_entries < 0 || num_bands < 1 || rows_per_band < 1 || dim < 1 ||
      ldx < dim || n_entries > INT32_MAX)
    return LSHRS_HOST_E_BADARG;
  Engine* e = static_cast<Engine*>(engine);
  std::lock_guard<std::mutex> run(e->run_mu);
  // entry = (row * 65536 + index of a 32-column word of the padded key row, mask of its flagged columns);
  // bands are padded to whole bytes: column c belongs to band c / (8 * band_bytes)
  const int band_cols = 8 * ((rows_per_band + 7) / 8);
  using Pair = PairRec;
  std::vector<Pair>& pairs = e->scratch_pairs;
  pairs.clear();
  for (int64_t i = 0; i < n_entries; ++i) {
    const int64_t row = entries[2 * i] >> 16;
    const int64_t word = entries[2 * i] & 0xFFFF;
    uint32_t mask = (uint32_t)entries[2 * i + 1];
    int last = -1;
    while (mask != 0u) {
      const int c = __builtin_ctz(mask);
      mask &= mask - 1u;
      const int b = (int)((32 * word + c) / band_cols);
      if (b != last && b < num_bands) pairs.push_back(Pair{row, b, (int32_t)i});
      last = b;
    }
  }
  // unique (band, row), sorted by band then row (a band of more than 32 columns spans several words)
  std::sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) {
    return a.band != b.band ? a.band < b.band : (a.row != b.row ? a.row < b.row : a.entry < b.entry);
  });
  pairs.erase(std::unique(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.band == b.band && a.row == b.row; }),
              pairs.end());
  const int64_t m = (int64_t)pairs.size();
  *n_pairs = m;
  if (m > out_cap) return LSHRS_HOST_E_BADARG;
  std::vector<int32_t>& idx = e->scratch_index;
  idx.resize((size_t)m);
  for (int64_t p = 0; p < m; ++p) {
    out_rows[p] = pairs[p].row;
    out_bands[p] = pairs[p].band;
    idx[p] = pairs[p].entry;   // the device staged one vector per entry: any entry of the row carries it
  }
  return run_job(e, planes, rows_per_band, dim, xstage, ldx, idx.data(), out_bands, m, out_keys, nullptr);
}

}  // extern "C"
