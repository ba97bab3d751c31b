// Diagnostic only: where one FPS round spends its cycles (s_memtime stamps), and the round time of alternative launch
// geometries.  The stamping lives HERE, as a probe policy handed to the kernel template of vision3d_amd/csrc/fps_kernel.h;
// the shipped library instantiates that template with FpsNoProbe only.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None \
//              tools/fps_probe.cpp -o tools/fps_probe.bin
#include "../vision3d_amd/csrc/runtime.hip"
#include "../vision3d_amd/csrc/fps.hip"
#include <cstdlib>
#include <vector>

__device__ unsigned long long g_fps_stamps[16];

struct StampProbe {
  unsigned long long prev, acc[8];
  __device__ __forceinline__ void begin() {
    prev = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 8; ++k) acc[k] = 0;
  }
  __device__ __forceinline__ void mark(int k) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    acc[k] += now - prev;
    prev = now;
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void flush() {
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (int k = 0; k < 8; ++k) g_fps_stamps[k] = acc[k];
  }
};

template <int T, int P, int MODE, class PROBE, int J>
static float launch(const float* pts, float* dist, int64_t* idx, int B, int N, int M, int lb, const std::vector<float>& d0) {
  hipMemcpy(dist, d0.data(), d0.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&v3d::fps_kernel<T, P, MODE, PROBE, J>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((v3d::fps_kernel<T, P, MODE, PROBE, J>), dim3(B), dim3(T), MODE == v3d::kFpsTable ? (size_t)T * P * 16 : 0, 0,
                     pts, dist, idx, N, M, lb, 0, M);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int T, int P, int MODE, int J = 0>
static void run(int B, int N, int M) {
  std::vector<float> h((size_t)B * N * 3);
  for (auto& v : h) v = (float)rand() / RAND_MAX;
  float *pts, *dist;
  int64_t* idx;
  hipMalloc(&pts, h.size() * 4);
  hipMalloc(&dist, (size_t)B * N * 4);
  hipMalloc(&idx, (size_t)B * M * 8);
  hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> d0((size_t)B * N, 1e10f);
  const int bs = v3d_fps_ref_block(N);
  int lb = 0;
  while ((1 << (lb + 1)) <= bs) ++lb;
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) best = std::min(best, launch<T, P, MODE, v3d::FpsNoProbe, J>(pts, dist, idx, B, N, M, lb, d0));
  const float ms = launch<T, P, MODE, StampProbe, J>(pts, dist, idx, B, N, M, lb, d0);
  unsigned long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fps_stamps), sizeof(st));
  printf("T=%d PPT=%d mode=%d B=%d N=%d M=%d: %.1f us, %.3f us/round (plain) | stamped %.1f us; s_memtime ticks/round:", T, P,
         MODE, B, N, M, best * 1e3, best * 1e3 / (M - 1), ms * 1e3);
  const char* names[7] = {"loop-top", "scan", "wave-max", "select+publish", "barrier", "slot-read+reduce", "broadcast"};
  for (int k = 0; k < 7; ++k) printf(" %s=%.0f", names[k], (double)st[k] / (M - 1));
  printf("\n");
  hipFree(pts);
  hipFree(dist);
  hipFree(idx);
}

int main() {
  using namespace v3d;
  run<512, 8, kFpsTable>(32, 4096, 1024);
  run<256, 16, kFpsTable, 1>(32, 4096, 1024);
  run<512, 2, kFpsTable>(16, 1024, 512);
  run<256, 4, kFpsTable, 1>(16, 1024, 512);
  run<512, 4, kFpsTable>(16, 2048, 512);
  run<1024, 8, kFpsTable>(8, 8192, 1024);
  run<1024, 16, kFpsRegs>(8, 16384, 512);
  run<1024, 1, kFpsStream>(2, 100000, 256);
  return 0;
}
