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

// Scratch of an entry point: the caller's workspace (the reference's ownership contract: the extension neither allocates
// nor frees, e.g. ops/ball_query.py:38 -- checked for size and 256-byte alignment) or, with workspace == NULL, a
// stream-ordered allocation that release() hands back on the same stream.
struct Scratch {
  char* base = nullptr;
  bool own = false;
  hipStream_t stream = nullptr;
  int acquire(const char* what, void* workspace, size_t workspace_bytes, size_t need, hipStream_t s) {
    stream = s;
    if (workspace != nullptr) {
      if (workspace_bytes < need || ((uintptr_t)workspace & 255) != 0) {
        set_last_error("%s: workspace needs %zu bytes, 256-B aligned (got %zu)", what, need, workspace_bytes);
        return V3D_ERR_WORKSPACE;
      }
      base = static_cast<char*>(workspace);
      return 0;
    }
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&base), need ? need : 256, s);
    if (e != hipSuccess) {
      set_last_error("%s: hipMallocAsync(%zu): %s", what, need, hipGetErrorString(e));
      return (int)e;
    }
    own = true;
    return 0;
  }
  int release(const char* what, int rc) {
    if (own) {
      hipError_t e = hipFreeAsync(base, stream);
      own = false;
      if (rc == 0 && e != hipSuccess) {
        set_last_error("%s: hipFreeAsync: %s", what, hipGetErrorString(e));
        rc = (int)e;
      }
    }
    return rc;
  }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Grid dims y/z are limited to 65535; x to 2^31-1.
constexpr int kMaxGridYZ = 65535;

// radius.hip: pack-mode kNN (k <= 64, at most 4096 clouds) on the radius search's cell list -- query.hip's knn_pack
// sends the clouds that are worth a grid there.
constexpr int kKnnCellListMaxK = 64;
constexpr int kKnnCellListMaxClouds = 4096;
int knn_pack_cell_list(const float* q_points, const float* s_points, const int64_t* q_lengths, const int64_t* s_lengths,
                       float* distances, int64_t* indices, int B, int64_t total_q, int64_t total_s, int k,
                       void* workspace, size_t workspace_bytes, int device, void* stream);
size_t knn_pack_cell_list_workspace_bytes(int B, int64_t total_s);


// XCD-aware workgroup placement.  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with a private
// 4 MiB L2.  All position tiles and channel tiles of one cloud re-read that cloud's source rows and indices, so the
// launch is a 1-D grid decoded such that every workgroup of a cloud (mode 1, B % 8 == 0) or of a (cloud, channel
// tile) pair (mode 2) carries the same id % 8: the rows are then fetched from HBM by ONE L2 instead of eight.
// Placement is a speed matter only; any mapping gives the same result.
struct GridMap {
  int X, Y, mode;
};
__device__ __forceinline__ void decode_block(const GridMap g, int& bx, int& by, int& bz) {
  const unsigned id = blockIdx.x;
  if (g.mode == 0) {
    bx = id % g.X;
    by = (id / g.X) % g.Y;
    bz = id / (g.X * g.Y);
  } else {
    const unsigned xcd = id & 7u, slot = id >> 3;
    if (g.mode == 1) {
      const unsigned per_b = g.X * g.Y;
      bz = (slot / per_b) * 8 + xcd;
      const unsigned rem = slot % per_b;
      bx = rem % g.X;
      by = rem / g.X;
    } else {
      const unsigned grp = (slot / g.X) * 8 + xcd;
      bx = slot % g.X;
      by = grp % g.Y;
      bz = grp / g.Y;
    }
  }
}
inline GridMap make_grid_map(int64_t X, int64_t Y, int64_t B) {
  GridMap g;
  g.X = (int)X;
  g.Y = (int)Y;
  g.mode = (B % 8 == 0) ? 1 : ((B * Y) % 8 == 0 ? 2 : 0);
  return g;
}

}  // namespace v3d

#endif
