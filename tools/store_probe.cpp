// store_probe -- which store stream does MI355X's memory system like?  (tuning aid for the group-gather forward kernel:
// its output (B, C, L) fp32 is written by workgroups that own CT channel rows x TILE bytes of one cloud.)
// build: hipcc --offload-arch=gfx950 -O3 -x hip -o tools/store_probe.bin tools/store_probe.cpp     run: tools/store_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Cfg {
  int B, C;
  long long L;       // floats per row
  int CT;            // channel rows per workgroup
  int tile_floats;   // floats of a row per workgroup (multiple of THREADS*4)
  int xcd;           // 1: all workgroups of a cloud on one XCD (id % 8 == b % 8)
  int nt;            // nontemporal stores
  int sync;          // __syncthreads between rows
  int order;         // 0: x fastest, then channel tile, then cloud;  1: channel tile fastest, then x, then cloud
};

template <int THREADS>
__global__ __launch_bounds__(THREADS) void store_kernel(float* __restrict__ out, Cfg c) {
  const int X = (int)(c.L / c.tile_floats), Y = c.C / c.CT;
  unsigned id = blockIdx.x;
  int bx, by, b;
  if (c.xcd) {
    const unsigned xcd = id & 7u, slot = id >> 3, per_b = X * Y;
    b = (slot / per_b) * 8 + xcd;
    id = slot % per_b;
  } else {
    b = id / (X * Y);
    id = id % (X * Y);
  }
  if (c.order == 0) { bx = id % X; by = id / X; } else { by = id % Y; bx = id / Y; }
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  const int R = c.tile_floats / (THREADS * 4);
  for (int ch = 0; ch < c.CT; ++ch) {
    float* row = out + ((long long)b * c.C + (long long)by * c.CT + ch) * c.L + (long long)bx * c.tile_floats;
    for (int r = 0; r < R; ++r) {
      f32x4* p = reinterpret_cast<f32x4*>(row + ((long long)r * THREADS + threadIdx.x) * 4);
      if (c.nt) __builtin_nontemporal_store(v, p); else *p = v;
    }
    if (c.sync) __syncthreads();
  }
}

// torch-like linear fill: block = 256 threads x 4 float4 at 4 KiB stride = 16 KiB contiguous
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ out, long long n4) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  const long long base = (long long)blockIdx.x * 1024;
  for (int j = 0; j < 4; ++j) {
    const long long i = base + j * 256 + threadIdx.x;
    if (i < n4) reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

static float median_ms(std::vector<float>& t) { std::sort(t.begin(), t.end()); return t[t.size() / 2]; }

int main() {
  const int B = 32, C = 64;
  const long long L = 131072;
  const size_t bytes = (size_t)B * C * L * 4;
  float* out;
  if (hipMalloc(&out, bytes) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_it = [&](auto launch) {
    std::vector<float> ts;
    for (int i = 0; i < 12; ++i) {
      hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (i >= 2) ts.push_back(ms);
    }
    return median_ms(ts);
  };
  float ms = time_it([&] { hipMemsetAsync(out, 0, bytes, 0); });
  printf("hipMemsetAsync                                   %7.1f us  %5.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
  ms = time_it([&] { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, out, (long long)(bytes / 16)); });
  printf("linear fill 16 KiB/block                         %7.1f us  %5.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
  const int cts[] = {1, 4, 16, 64};
  const int tiles[] = {4096, 8192, 32768};
  for (int threads : {256, 512})
    for (int tile : tiles)
      for (int ct : cts)
        for (int xcd = 0; xcd < 2; ++xcd)
          for (int order = 0; order < 2; ++order)
            for (int nt = 0; nt < 2; ++nt) {
              if (tile < threads * 4) continue;
              if (order == 1 && ct == 64) continue;
              Cfg c{B, C, L, ct, tile, xcd, nt, 1, order};
              const unsigned grid = (unsigned)(B * (C / ct) * (L / tile));
              ms = time_it([&] {
                if (threads == 256) hipLaunchKernelGGL(store_kernel<256>, dim3(grid), dim3(256), 0, 0, out, c);
                else hipLaunchKernelGGL(store_kernel<512>, dim3(grid), dim3(512), 0, 0, out, c);
              });
              printf("thr %3d tile %6d B ct %2d xcd %d order %d nt %d  grid %6u  %7.1f us  %5.2f TB/s\n", threads, tile * 4, ct, xcd,
                     order, nt, grid, ms * 1e3, bytes / ms / 1e9);
              fflush(stdout);
            }
  hipFree(out);
  return 0;
}
