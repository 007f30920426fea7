// sdma_probe.hip — does a device -> pinned-host copy that runs BESIDE the split-precision signature pass slow it down,
// and does that depend on who moves the bytes?  Runs 8 launches of 262 144 x 768 rows through liblshrs_hip.so
//   mode 0  alone
//   mode 1  while a host thread keeps hsa_amd_memory_async_copy (copy engine, SDMA) transfers of 3 MB in flight
//   mode 2  while a host thread keeps hipMemcpyAsync transfers of 3 MB in flight on a second stream
//   mode 3  as 2, on a high-priority stream
//   mode 4  as 2, each copy behind a tiny kernel on the same stream (what the pipeline's side stream looks like)
// and prints the time of the 8 launches (HIP events).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/sdma_probe.hip -o tools/sdma_probe.bin -lhsa-runtime64 -ldl && tools/sdma_probe.bin
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

__global__ void fill_kernel(float* x, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = (uint32_t)i * 2654435761u + seed;
  s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
  x[i] = ((float)(s & 0xFFFFFF) / 8388608.0f - 1.0f) * 1.7f;   // roughly unit variance
}

static hsa_agent_t g_gpu{0}, g_cpu{0};
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
  if (t == HSA_DEVICE_TYPE_CPU && g_cpu.handle == 0) g_cpu = a;
  return HSA_STATUS_SUCCESS;
}

typedef int64_t (*ws_bytes_fn)(int32_t, int32_t, int32_t);
typedef int (*pack_fn)(const float*, int32_t, int32_t, int32_t, void*, void*);
typedef int (*split_fn)(const float*, int64_t, int64_t, const void*, int32_t, int32_t, int32_t, uint8_t*, int64_t*, int32_t,
                        int32_t*, float, uint8_t*, int64_t*, int32_t, int32_t*, float, void*);

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "lshrs_amd/csrc/liblshrs_hip.so";
  void* lib = dlopen(libpath, RTLD_NOW);
  if (!lib) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 1; }
  auto ws_bytes = (ws_bytes_fn)dlsym(lib, "lshrs_sig_workspace_bytes");
  auto pack = (pack_fn)dlsym(lib, "lshrs_sig_pack_projections");
  auto split = (split_fn)dlsym(lib, "lshrs_sig_hash_batch_split_f32");
  const int nb = 16, r = 16, dim = 768;
  const int64_t n = 262144;
  const int launches = 8;
  CK(hipSetDevice(0));
  float *X, *P;
  CK(hipMalloc(&X, sizeof(float) * n * dim));
  CK(hipMalloc(&P, sizeof(float) * nb * r * dim));
  fill_kernel<<<(n * dim + 255) / 256, 256>>>(X, (size_t)n * dim, 1u);
  fill_kernel<<<(nb * r * dim + 255) / 256, 256>>>(P, (size_t)nb * r * dim, 7u);
  void* ws;
  CK(hipMalloc(&ws, ws_bytes(nb, r, dim)));
  if (pack(P, nb, r, dim, ws, nullptr) != 0) { fprintf(stderr, "pack failed\n"); return 1; }
  uint8_t* keys;
  CK(hipMalloc(&keys, n * 32));
  const int tie_cap = 100000, flag_cap = (int)(n / 4 + 4096);
  int64_t *tie_list, *flag_list;
  int32_t* counts;
  CK(hipMalloc(&tie_list, 16 * (size_t)tie_cap));
  CK(hipMalloc(&flag_list, 8 * (size_t)flag_cap));
  CK(hipMalloc(&counts, 8 * launches));
  const size_t copy_bytes = 3u << 20;
  char *dsrc, *hdst;
  CK(hipMalloc(&dsrc, copy_bytes));
  CK(hipHostMalloc(&hdst, copy_bytes, hipHostMallocDefault));
  CK(hipDeviceSynchronize());

  if (hsa_init() != HSA_STATUS_SUCCESS) { fprintf(stderr, "hsa_init failed\n"); return 1; }
  hsa_iterate_agents(agent_cb, nullptr);
  hsa_signal_t sig;
  hsa_signal_create(1, 0, nullptr, &sig);
  uint32_t engines = 0;
  hsa_status_t es = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &engines);
  printf("copy engines free for gpu -> host: status %d mask 0x%x\n", (int)es, engines);

  hipStream_t main_s, side, side_hi;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  {
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&side_hi, hipStreamNonBlocking, hi));
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const float u = 1.0f / 16777216.0f;

  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
      std::atomic<bool> stop{false};
      std::atomic<long> copies{0};
      std::thread bg;
      if (mode == 1)
        bg = std::thread([&] {
          while (!stop.load()) {
            hsa_signal_store_relaxed(sig, 1);
            hsa_status_t st = hsa_amd_memory_async_copy(hdst, g_cpu, dsrc, g_gpu, copy_bytes, 0, nullptr, sig);
            if (st != HSA_STATUS_SUCCESS) { fprintf(stderr, "async copy failed %d\n", (int)st); break; }
            hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE);
            copies.fetch_add(1);
          }
        });
      else if (mode >= 2)
        bg = std::thread([&, mode] {
          (void)hipSetDevice(0);
          hipStream_t ss = mode == 3 ? side_hi : side;
          while (!stop.load()) {
            if (mode == 4) fill_kernel<<<1, 64, 0, ss>>>(reinterpret_cast<float*>(dsrc), 64, 3u);
            (void)hipMemcpyAsync(hdst, dsrc, copy_bytes, hipMemcpyDeviceToHost, ss);
            (void)hipStreamSynchronize(ss);
            copies.fetch_add(1);
          }
        });
      CK(hipMemsetAsync(counts, 0, 8 * launches, main_s));
      CK(hipEventRecord(a, main_s));
      for (int l = 0; l < launches; ++l) {
        int rc = split(X, n, dim, ws, nb, r, dim, keys, tie_list, tie_cap, counts + 2 * l, 8 * u, nullptr, flag_list, flag_cap,
                       counts + 2 * l + 1, 64 * u, main_s);
        if (rc != 0) { fprintf(stderr, "split rc %d\n", rc); return 1; }
      }
      CK(hipEventRecord(b, main_s));
      CK(hipEventSynchronize(b));
      stop.store(true);
      if (bg.joinable()) bg.join();
      float ms = 0;
      CK(hipEventElapsedTime(&ms, a, b));
      printf("rep %d mode %d (%s): %d launches of %ld rows in %.3f ms = %.3f ms per launch; %ld copies of 3 MB beside them (%.1f GB/s)\n",
             rep, mode, mode == 0 ? "alone" : (mode == 1 ? "hsa async copy" : (mode == 2 ? "hipMemcpyAsync" : (mode == 3 ? "hipMemcpyAsync, high-priority stream" : "kernel + hipMemcpyAsync"))), launches, (long)n, ms,
             ms / launches, copies.load(), copies.load() * (double)copy_bytes / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
  return 0;
}
