// Diagnostic only: where one FPS round spends its cycles (s_memtime stamps, never in the shipped library).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DV3D_FPS_STAMPS tools/fps_probe.cpp -o /tmp/fps_probe
#include "../vision3d_amd/csrc/runtime.hip"
#include "../vision3d_amd/csrc/fps.hip"
#include <vector>
#include <cstdlib>
template <int T, int P, int MODE>
static void run(int B, int N, int M) {
  std::vector<float> h((size_t)B * N * 3);
  for (auto& v : h) v = (float)rand() / RAND_MAX;
  float *pts, *dist; int64_t* idx;
  hipMalloc(&pts, h.size() * 4); hipMalloc(&dist, (size_t)B * N * 4); hipMalloc(&idx, (size_t)B * M * 8);
  hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> d0((size_t)B * N, 1e10f);
  const int bs = v3d_fps_ref_block(N); int lb = 0; while ((1 << (lb + 1)) <= bs) ++lb;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dist, d0.data(), d0.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((v3d::fps_kernel<T, P, MODE>), dim3(B), dim3(T), MODE == 0 ? (size_t)N * 16 : 0, 0, pts, dist, idx, N, M, lb);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(v3d::g_fps_stamps), sizeof(st));
    printf("T=%d PPT=%d mode=%d N=%d M=%d: %.1f us total, %.3f us/round | cycles/round:", T, P, MODE, N, M, ms * 1e3, ms * 1e3 / (M - 1));
    const char* names[7] = {"loop-top", "scan", "wave-max", "publish", "barrier", "key-read+decode", "xyz-read"};
    for (int k = 0; k < 7; ++k) printf(" %s=%.0f", names[k], (double)st[k] / (M - 1));
    printf("\n");
  }
  hipFree(pts); hipFree(dist); hipFree(idx);
}
int main() {
  run<512, 8, 0>(32, 4096, 1024);
  run<1024, 4, 0>(32, 4096, 1024);
  run<512, 8, 1>(32, 4096, 1024);
  run<256, 16, 0>(32, 4096, 1024);
  run<512, 2, 0>(16, 1024, 512);
  run<256, 4, 0>(16, 1024, 512);
  return 0;
}
