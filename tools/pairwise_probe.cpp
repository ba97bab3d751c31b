// Diagnostic only: ablations of the pairwise_distance kernel (no stores / no MFMA), never in the shipped library.
#include "../vision3d_amd/csrc/runtime.hip"
#include "../vision3d_amd/csrc/pairwise.hip"
#include <vector>
#include <cstdlib>
static void run(int batch, int N, int M, int C) {
  size_t nx = (size_t)batch * N * C, ny = (size_t)batch * M * C, no = (size_t)batch * N * M;
  std::vector<float> h(nx > ny ? nx : ny);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *x, *y, *o;
  hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&o, no * 4);
  hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(a);
    v3d_pairwise_distance_f32(x, y, o, batch, N, M, C, 0, 0.f, 0, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  printf("batch %d N %d M %d C %d: %.1f us  %.1f TFLOP/s  out %.0f GB/s\n", batch, N, M, C, best * 1e3,
         2.0 * batch * N * M * C / (best * 1e-3) / 1e12, no * 4.0 / (best * 1e-3) / 1e9);
  hipFree(x); hipFree(y); hipFree(o);
}
int main() { run(32, 1024, 1024, 64); run(1, 8192, 8192, 256); run(1, 4096, 4096, 4096); return 0; }
