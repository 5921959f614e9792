// tl_runtime.hip — device / memory / stream entry points of the C-ABI. They stand in for the raw
// cudaSetDevice / cudaMalloc / cudaMemcpy / cudaFree / cudaStream* calls the reference's host files make
// inline (model.cpp, gpu_initialization.cpp, weight_management.cpp, kv_cache.cpp, api.cpp).
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <utility>

#include "tl_internal.h"

static thread_local char g_err[1024] = "";

void tl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char* tl_last_error(void) { return g_err; }

extern "C" int tl_device_count(int* count_out) {
  TL_REQUIRE(count_out, "tl_device_count: null argument");
  TL_HIP(hipGetDeviceCount(count_out));
  return 0;
}

extern "C" int tl_init(int device_ordinal) {
  int n = 0;
  TL_HIP(hipGetDeviceCount(&n));
  TL_REQUIRE(device_ordinal >= 0 && device_ordinal < n, "tl_init: device %d not in [0,%d)", device_ordinal, n);
  TL_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  TL_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  TL_REQUIRE(prop.warpSize == 64, "tl_init: wavefront size %d, kernels are written for wave64 (gfx950)", prop.warpSize);
  TL_HIP(hipFree(nullptr));  // force context creation here rather than inside the first timed call
  return tl_ops_init_device(device_ordinal);
}

extern "C" int tl_malloc(void** dev_out, size_t bytes) {
  TL_REQUIRE(dev_out, "tl_malloc: null argument");
  TL_HIP(hipMalloc(dev_out, bytes ? bytes : 1));
  return 0;
}
extern "C" int tl_free(void* dev) {
  if (dev) TL_HIP(hipFree(dev));
  return 0;
}
extern "C" int tl_memset(void* dev, int value, size_t bytes, void* stream) {
  TL_HIP(hipMemsetAsync(dev, value, bytes, static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" int tl_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
  TL_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
  TL_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // host buffer is borrowed for the call only
  return 0;
}
extern "C" int tl_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
  TL_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
  TL_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" int tl_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
  TL_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" int tl_stream_create(void** stream_out) {
  TL_REQUIRE(stream_out, "tl_stream_create: null argument");
  hipStream_t s;
  TL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream_out = s;
  return 0;
}
extern "C" int tl_stream_destroy(void* stream) {
  if (stream) TL_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" int tl_stream_sync(void* stream) {
  TL_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" int tl_device_sync(void) {
  TL_HIP(hipDeviceSynchronize());
  return 0;
}

// ---- workspaces of the operator-level entry points ------------------------------------------------------------------------
// tl_attention (split partials), tl_matmul_q (activation planes) and tl_sample_top_k_top_p (probabilities, parameters)
// need device memory beyond their arguments. One lazily grown allocation per (device, slot), handed out under a mutex;
// growing synchronises the device first (an earlier launch may still read the old block). Calls that share a slot must
// not overlap on one device: one host thread per stream, as in the reference. tl_shutdown() frees them.
namespace {
std::mutex g_scratch_mu;
std::map<std::pair<int, int>, std::pair<void*, size_t>> g_scratch;
}  // namespace

int tl_scratch(int slot, size_t bytes, void** out) {
  int dev = 0;
  TL_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto& e = g_scratch[{dev, slot}];
  if (e.second < bytes) {
    TL_HIP(hipDeviceSynchronize());
    if (e.first) TL_HIP(hipFree(e.first));
    e = {nullptr, 0};
    void* p = nullptr;
    TL_HIP(hipMalloc(&p, bytes));
    TL_HIP(hipMemset(p, 0, bytes));
    e = {p, bytes};
  }
  *out = e.first;
  return 0;
}

extern "C" int tl_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int cur = 0;
  TL_HIP(hipGetDevice(&cur));
  for (auto& kv : g_scratch) {
    if (!kv.second.first) continue;
    TL_HIP(hipSetDevice(kv.first.first));
    TL_HIP(hipDeviceSynchronize());
    TL_HIP(hipFree(kv.second.first));
  }
  g_scratch.clear();
  TL_HIP(hipSetDevice(cur));
  return 0;
}
