// Library-level entry points and the thread-local error channel of libdyolo.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "dyolo.h"

namespace dy {

static thread_local char g_err[512] = {0};
static thread_local int g_stats = 0;              // dy_conv_stats_written()
static thread_local const char* g_kernel = "";  // name the last launching call passed to check_launch (static storage)

void note_stats(int written) { g_stats = written; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Launch errors surface here; hipGetLastError also clears the sticky flag so that one
// failed call does not poison the next.
int check_launch(const char* what) {
  g_kernel = what;
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return DY_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return DY_ERR_LAUNCH;
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* p, long long words) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256) p[i] = 0u;
}

void zero_async(void* p, size_t bytes, hipStream_t stream) {
  const long long words = (long long)(bytes / 4);
  if (words <= 0) return;
  const long long blocks = (words + 255) / 256;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, stream, reinterpret_cast<unsigned*>(p), words);
}

}  // namespace dy

extern "C" int32_t dy_version(void) { return (DYOLO_VERSION_MAJOR << 16) | DYOLO_VERSION_MINOR; }

extern "C" const char* dy_last_error_string(void) { return dy::g_err; }

extern "C" const char* dy_last_kernel_name(void) { return dy::g_kernel; }

extern "C" int32_t dy_conv_stats_written(void) { return dy::g_stats; }

extern "C" int32_t dy_dtype_size(int32_t dtype) {
  switch (dtype) {
    case DY_BF16:
    case DY_F16:
      return 2;
    case DY_F32:
    case DY_F16X2:
      return 4;
    case DY_FP8:
      return 1;
    default:
      return 0;
  }
}
