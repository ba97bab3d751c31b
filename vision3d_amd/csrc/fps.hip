// fps.hip -- iterative furthest point sampling, one workgroup per cloud: the C-ABI entry point and the launch table.
// The kernel and its design notes live in fps_kernel.h.
#include <math.h>

#include <mutex>

#include "fps_kernel.h"

namespace v3d {

static int ilog2_floor(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) ++l;
  return l;
}

}  // namespace v3d

extern "C" {

int v3d_fps_ref_block(int num_points) {
  // opt_n_threads, furthest_point_sample_cuda.cuh:9-12 -- same double expression, same truncation
  if (num_points <= 0) return 1;
  int pow_2 = static_cast<int>(std::log(static_cast<double>(num_points)) / std::log(2.0));
  int b = 1 << pow_2;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return b;
}

int v3d_fps_f32(const float* points, float* distances, int64_t* indices, int B, int N, int M, int device,
                void* stream_) {
  return v3d_fps_resume_f32(points, distances, indices, B, N, M, 0, M, device, stream_);
}

int v3d_fps_resume_f32(const float* points, float* distances, int64_t* indices, int B, int N, int M_total, int first,
                       int last, int device, void* stream_) {
  using namespace v3d;
  const char* what = "furthest_point_sample";
  V3D_REQUIRE(B >= 0 && N >= 0 && M_total >= 0, what, "negative dimension");
  V3D_REQUIRE(first >= 0 && first % 64 == 0, what, "a resumed sampling starts at a multiple of 64 rounds");
  V3D_REQUIRE(last <= M_total, what, "rounds beyond the row of samples");
  const int M = last;  // this launch: rounds [first, last) of rows of M_total samples
  if (B == 0 || M <= 0 || first >= M) return 0;
  V3D_REQUIRE(N > 0, what, "cannot sample from an empty cloud");
  V3D_REQUIRE(points && distances && indices, what, "null pointer");
  if ((int64_t)N * 3 > (int64_t)INT32_MAX) {
    set_last_error("%s: N*3 must fit in 31 bits", what);
    return V3D_ERR_SIZE_LIMIT;
  }
  DeviceGuard guard(device);
  if (guard.rc) return guard.rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int bs = v3d_fps_ref_block(N);
  const int lb = ilog2_floor(bs);
#define V3D_FPS_J(T_, P_, M_, J_)                                                                          \
  hipLaunchKernelGGL((fps_kernel<T_, P_, M_, FpsNoProbe, J_>), dim3(B), dim3(T_),                          \
                     (M_) == kFpsTable ? (size_t)(T_) * (P_) * sizeof(float4) : 0, stream, points, distances, \
                     indices, N, M, lb, first, M_total)
#define V3D_FPS(T_, P_, M_) V3D_FPS_J(T_, P_, M_, 0)
  // THREADS is a multiple of the reference block `bs`, or bs / 2^J (see fps_kernel.h): bs <= 512 is a power of two.
  // Geometries from tools/fps_probe.cpp: a round costs about (instructions issued per SIMD) x 5 cycles, so small clouds
  // run one wave per SIMD (every wave pays the ~60 instructions of the reduction chain, the scan is the smaller part).
  if (N <= 64) V3D_FPS(64, 1, kFpsTable);          // bs <= 64
  else if (N < 128) V3D_FPS(64, 2, kFpsTable);     // bs == 64
  else if (N < 256) V3D_FPS(128, 2, kFpsTable);    // bs == 128
  else if (N < 512) V3D_FPS(256, 2, kFpsTable);    // bs == 256
  else if (N <= 1024) V3D_FPS_J(256, 4, kFpsTable, 1);  // bs == 512 from here on; 256 threads = bs / 2
  else if (N <= 2048) V3D_FPS(512, 4, kFpsTable);
  else if (N <= 4096) V3D_FPS(512, 8, kFpsTable);
  else if (N <= 8192) {
    // 128 KiB of dynamic LDS needs the opt-in, once per device; the per-device flags are the library's only state on
    // this path and are read and written under a mutex (calls may come from several threads)
    static std::mutex attr_mutex;
    static bool attr_set[64] = {};
    {
      std::lock_guard<std::mutex> lock(attr_mutex);
      if (device < 0 || device >= 64 || !attr_set[device]) {
        V3D_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_kernel<1024, 8, kFpsTable>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * (int)sizeof(float4)),
                    what);
        if (device >= 0 && device < 64) attr_set[device] = true;
      }
    }
    V3D_FPS(1024, 8, kFpsTable);
  } else if (N <= 16384) V3D_FPS(1024, 16, kFpsRegs);
  else V3D_FPS(1024, 1, kFpsStream);
#undef V3D_FPS
#undef V3D_FPS_J
  return launch_status(what);
}

}  // extern "C"
