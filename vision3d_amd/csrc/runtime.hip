// runtime.hip -- library identification and the thread-local error string of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace v3d {

static thread_local char g_last_error[512] = "";

void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

}  // namespace v3d

extern "C" {

int v3d_version(void) { return 10000 * 0 + 100 * 1 + 0; }

const char* v3d_last_error_string(void) { return v3d::g_last_error; }

const char* v3d_error_name(int code) {
  switch (code) {
    case V3D_OK: return "V3D_OK";
    case V3D_ERR_INVALID_ARGUMENT: return "V3D_ERR_INVALID_ARGUMENT";
    case V3D_ERR_K_TOO_LARGE: return "V3D_ERR_K_TOO_LARGE";
    case V3D_ERR_WORKSPACE: return "V3D_ERR_WORKSPACE";
    case V3D_ERR_SIZE_LIMIT: return "V3D_ERR_SIZE_LIMIT";
    default: break;
  }
  if (code > 0) return hipGetErrorName((hipError_t)code);
  return "V3D_ERR_UNKNOWN";
}

int v3d_device_count(int* count) {
  if (!count) return V3D_ERR_INVALID_ARGUMENT;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    v3d::set_last_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // extern "C"
