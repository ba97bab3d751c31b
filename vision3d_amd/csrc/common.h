// common.h -- host-side plumbing shared by every libv3d_hip translation unit (gfx950 only).
#ifndef V3D_CSRC_COMMON_H_
#define V3D_CSRC_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/v3d_hip.h"

namespace v3d {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_last_error(const char* fmt, ...);

// RAII device switch: the reference wraps every launcher in at::cuda::CUDAGuard (ball_query_cuda.cu:16).
struct DeviceGuard {
  int prev = -1;
  int rc = 0;
  explicit DeviceGuard(int device) {
    rc = (int)hipGetDevice(&prev);
    if (rc == 0 && prev != device) rc = (int)hipSetDevice(device);
    if (rc != 0) set_last_error("cannot select device %d: %s", device, hipGetErrorString((hipError_t)rc));
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();  // == AT_CUDA_CHECK(cudaGetLastError()), ball_query_cuda.cu:38
  if (e != hipSuccess) {
    set_last_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

inline int invalid(const char* what, const char* why) {
  set_last_error("%s: %s", what, why);
  return V3D_ERR_INVALID_ARGUMENT;
}

#define V3D_REQUIRE(cond, what, why) \
  do {                               \
    if (!(cond)) return ::v3d::invalid(what, why); \
  } while (0)

#define V3D_HIP_TRY(expr, what)                                                   \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      ::v3d::set_last_error("%s: %s: %s", what, #expr, hipGetErrorString(e__));   \
      return (int)e__;                                                            \
    }                                                                             \
  } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Grid dims y/z are limited to 65535; x to 2^31-1.
constexpr int kMaxGridYZ = 65535;

}  // namespace v3d

#endif
