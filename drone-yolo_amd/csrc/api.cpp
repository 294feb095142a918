// Library-level entry points and the thread-local error channel of libdyolo.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "dyolo.h"

namespace dy {

static thread_local char g_err[512] = {0};
static thread_local const char* g_kernel = "";  // name the last launching call passed to check_launch (static storage)

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

}  // namespace dy

extern "C" int32_t dy_version(void) { return (DYOLO_VERSION_MAJOR << 16) | DYOLO_VERSION_MINOR; }

extern "C" const char* dy_last_error_string(void) { return dy::g_err; }

extern "C" const char* dy_last_kernel_name(void) { return dy::g_kernel; }

extern "C" int32_t dy_dtype_size(int32_t dtype) {
  switch (dtype) {
    case DY_BF16:
    case DY_F16:
      return 2;
    case DY_F32:
      return 4;
    case DY_FP8:
      return 1;
    default:
      return 0;
  }
}
