// host_tiebreak.cpp — host half of the signature tie-break (no GPU code).
//
// For the handful of (row, band) pairs the kernel reports as ties, the band key must be what the
// REFERENCE computes on this host: `projection @ vector` (lshrs/hash/lsh.py:200), which NumPy resolves
// to one cblas_sgemv call (numpy/_core/src/umath/matmul.c.src, gemv branch):
//     cblas_sgemv(CblasColMajor, CblasTrans, dim, rows, 1.0f, P_band, dim, x, 1, 0.0f, y, 1)
// This file issues exactly that call — through the function pointer of the BLAS library NumPy itself
// has loaded, handed in by the Python side — in a native loop over the pairs (NumPy's per-call
// dispatch costs more than the 16x768 sgemv itself), then applies `> 0` (lsh.py:204) and the
// LSB-first pack (lsh.py:208).  The Python side verifies bit-for-bit against `P @ x` before trusting it.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "lshrs_hip.h"

namespace {

constexpr int kColMajor = 102;  // CBLAS_ORDER::CblasColMajor
constexpr int kTrans = 112;     // CBLAS_TRANSPOSE::CblasTrans

typedef void (*sgemv_lp64_fn)(int, int, int, int, float, const float*, int, const float*, int, float, float*, int);
typedef void (*sgemv_ilp64_fn)(int, int, int64_t, int64_t, float, const float*, int64_t, const float*, int64_t, float,
                               float*, int64_t);

typedef int (*set_threads_local_fn)(int);

struct Job {
  void* fn;
  set_threads_local_fn set_local;  // openblas_set_num_threads_local, or NULL
  int ilp64;
  const float* const* planes;
  int rows, dim, bb;
  const float* xrows;
  const int64_t* xindex;
  const int32_t* bands;
  uint8_t* patch;
  float* y_out;
};

void run_range(const Job& j, int64_t lo, int64_t hi, bool worker) {
  // Inside a worker thread keep the BLAS call on this thread: OpenBLAS would otherwise route a
  // 12k-element sgemv through its global thread server, serialising the workers.  (Per output the
  // summation order is the same either way; the Python side checks that bit for bit.)
  int previous = -1;
  if (worker && j.set_local != nullptr) previous = j.set_local(1);
  std::vector<float> ybuf((size_t)j.rows);
  for (int64_t t = lo; t < hi; ++t) {
    const float* plane = j.planes[j.bands[t]];
    const float* x = j.xrows + j.xindex[t] * (int64_t)j.dim;
    float* y = j.y_out != nullptr ? j.y_out + t * (int64_t)j.rows : ybuf.data();
    if (j.ilp64)
      reinterpret_cast<sgemv_ilp64_fn>(j.fn)(kColMajor, kTrans, j.dim, j.rows, 1.0f, plane, j.dim, x, 1, 0.0f, y, 1);
    else
      reinterpret_cast<sgemv_lp64_fn>(j.fn)(kColMajor, kTrans, j.dim, j.rows, 1.0f, plane, j.dim, x, 1, 0.0f, y, 1);
    uint8_t* dst = j.patch + t * (int64_t)j.bb;
    memset(dst, 0, (size_t)j.bb);
    for (int r = 0; r < j.rows; ++r)
      if (y[r] > 0.0f) dst[r >> 3] |= (uint8_t)(1u << (r & 7));
  }
  if (worker && j.set_local != nullptr && previous >= 0) j.set_local(previous);
}

}  // namespace

extern "C" int lshrs_host_band_keys_f32(void* cblas_sgemv, int32_t ilp64, void* set_num_threads_local,
                                        const float* const* planes, int32_t rows_per_band,
                                        int32_t dim, const float* xrows, const int64_t* xindex, const int32_t* bands,
                                        int64_t m, uint8_t* patch, float* y_out, int32_t threads) {
  if (m == 0) return 0;
  if (cblas_sgemv == nullptr || planes == nullptr || xrows == nullptr || xindex == nullptr || bands == nullptr ||
      patch == nullptr || rows_per_band <= 0 || dim <= 0 || m < 0)
    return LSHRS_E_BADARG;
  Job j{cblas_sgemv, reinterpret_cast<set_threads_local_fn>(set_num_threads_local), ilp64, planes, rows_per_band, dim, (rows_per_band + 7) / 8, xrows, xindex, bands, patch, y_out};
  int nt = std::max(1, std::min<int>(threads, 64));
  nt = (int)std::min<int64_t>(nt, (m + 255) / 256);  // not worth a thread for fewer than ~256 pairs
  if (nt <= 1) {
    run_range(j, 0, m, false);
    return 0;
  }
  std::vector<std::thread> pool;
  const int64_t per = (m + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int64_t lo = t * per, hi = std::min<int64_t>(m, lo + per);
    if (lo >= hi) break;
    pool.emplace_back(run_range, std::cref(j), lo, hi, true);
  }
  for (auto& th : pool) th.join();
  return 0;
}
