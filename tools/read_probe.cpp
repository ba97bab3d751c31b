// Diagnostic only: what a read-only stream reaches on this chip, for the access patterns of the backward kernels.
//   A: grid-stride streaming read (16-byte loads, U in flight per thread), any number of workgroups
//   B: the segment-sum pattern: one 1024-thread workgroup per (cloud, channel tile), per chunk ROWS rows x 8 KiB, rows
//      512 KiB apart, two register sets, optional workgroup barrier per chunk
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/read_probe.cpp -o tools/read_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ x, size_t nvec, unsigned* out) {
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride < nvec ? i + u * stride : i];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int ROWS, bool BARRIER, int DEPTH>
__global__ __launch_bounds__(1024) void tile_read(const uint4* __restrict__ x, int nchunks, size_t row_vecs,
                                                  unsigned* out) {
  // workgroup w owns rows w*ROWS .. +ROWS-1; a chunk is 2048 columns = 512 uint4 per row
  constexpr int kVecPerRow = 512;
  constexpr int kLoads = ROWS * kVecPerRow / 1024;  // per thread and chunk
  static_assert(kLoads >= 1, "");
  const int tid = threadIdx.x;
  const int part = tid % (ROWS / kLoads), grp = tid / (ROWS / kLoads);
  const uint4* base = x + (size_t)blockIdx.x * ROWS * row_vecs;
  unsigned acc = 0;
  uint4 r[DEPTH][kLoads];
  for (int k = -DEPTH; k < nchunks; k += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      if (k >= 0) {
        if (BARRIER) __syncthreads();
#pragma unroll
        for (int i = 0; i < kLoads; ++i) acc ^= r[j][i].x ^ r[j][i].y ^ r[j][i].z ^ r[j][i].w;
        if (BARRIER) __syncthreads();
      }
      const int kk = k + j + DEPTH < nchunks ? k + j + DEPTH : nchunks - 1;
#pragma unroll
      for (int i = 0; i < kLoads; ++i)
        r[j][i] = base[(size_t)(part * kLoads + i) * row_vecs + (size_t)kk * kVecPerRow + grp];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t bytes = 1ull << 30;
  uint4* x;
  unsigned* out;
  hipMalloc(&x, bytes);
  hipMalloc(&out, 4);
  hipMemset(x, 1, bytes);
  const size_t nvec = bytes / 16;
  for (int wgs : {1024, 4096, 16384}) {
    float ms = time_ms([&] { hipLaunchKernelGGL(stream_read<4>, dim3(wgs), dim3(256), 0, 0, x, nvec, out); });
    printf("A stream_read U=4 wgs=%d: %.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(stream_read<8>, dim3(wgs), dim3(256), 0, 0, x, nvec, out); });
    printf("A stream_read U=8 wgs=%d: %.1f us  %.2f TB/s\n", wgs, ms * 1e3, bytes / ms / 1e9);
  }
  // B: 1 GiB = 2048 rows of 512 KiB (131072 columns): ROWS rows per workgroup, 64 chunks
  const size_t row_vecs = 131072 / 4;
#define RUN_B(ROWS, BAR, DEPTH)                                                                                    \
  {                                                                                                                 \
    float ms = time_ms([&] {                                                                                        \
      hipLaunchKernelGGL((tile_read<ROWS, BAR, DEPTH>), dim3(2048 / ROWS), dim3(1024), lds, 0, x, 64, row_vecs, out); \
    });                                                                                                             \
    printf("B tile_read rows=%d barrier=%d depth=%d wgs=%d lds=%zu: %.1f us  %.2f TB/s\n", ROWS, (int)BAR, DEPTH,  \
           2048 / ROWS, lds, ms * 1e3, bytes / ms / 1e9);                                                           \
  }
  size_t lds = 0;
  RUN_B(4, true, 2)
  RUN_B(4, false, 2)
  RUN_B(4, true, 4)
  RUN_B(4, false, 4)
  RUN_B(8, true, 2)
  RUN_B(8, false, 2)
  RUN_B(8, true, 4)
  RUN_B(2, true, 2)
  RUN_B(2, true, 4)
  lds = 100 * 1024;  // one workgroup per CU, as the segment-sum kernel's acc + tile force
  hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_read<4, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_read<4, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_read<8, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_read<8, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  RUN_B(4, true, 2)
  RUN_B(4, true, 4)
  RUN_B(8, true, 2)
  RUN_B(8, true, 4)
  hipFree(x);
  hipFree(out);
  return 0;
}
