// tl_runtime.hip — device / memory / stream entry points of the C-ABI. They stand in for the raw
// cudaSetDevice / cudaMalloc / cudaMemcpy / cudaFree / cudaStream* calls the reference's host files make
// inline (model.cpp, gpu_initialization.cpp, weight_management.cpp, kv_cache.cpp, api.cpp).
#include <stdarg.h>
#include <stdio.h>

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
  return 0;
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
