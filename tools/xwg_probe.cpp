// Diagnostic only (not part of the shipped library): the round trip of a flag between TWO WORKGROUPS through global memory
// -- what a furthest-point-sampling round would pay if one cloud were split over several CUs (fps_kernel row).
// Workgroup 0 and workgroup `partner` bounce a counter REP times: A waits for 2i, writes 2i+1; B waits for 2i+1, writes
// 2i+2 (release store / acquire load, agent scope; one lane per workgroup).  Workgroups are dealt round-robin over the 8
// XCDs (id % 8), so partner 8 shares workgroup 0's XCD and L2, partner 1 sits on the next XCD.  Every spin is bounded: a
// partner that never answers makes both sides give up (reported as a failure), so the grid always drains.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xwg_probe.cpp -o tools/xwg_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP 2000
#define SPIN_LIMIT (1 << 22)

__global__ void k_pingpong(unsigned* flag, unsigned long long* ticks, unsigned long long* clocks, int* failed, int partner) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)partner ? 1 : -1);
  if (me < 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long c0 = wall_clock64();
  bool ok = true;
  for (unsigned i = 0; i < REP && ok; ++i) {
    const unsigned want = 2 * i + (unsigned)me;
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
      if (++spins > SPIN_LIMIT || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        ok = false;
        break;
      }
    }
    if (!ok) break;
    __hip_atomic_store(flag, want + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!ok) __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (me == 0) {
    ticks[0] = __builtin_amdgcn_s_memtime() - t0;
    clocks[0] = wall_clock64() - c0;
  }
}

int main() {
  unsigned* flag;
  unsigned long long *ticks, *clocks;
  int* failed;
  hipMalloc(&flag, 256);
  hipMalloc(&ticks, 256);
  hipMalloc(&clocks, 256);
  hipMalloc(&failed, 256);
  int rate_khz = 0;
  hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  for (int partner : {8, 1, 16, 4}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(flag, 0, 256);
      hipMemset(failed, 0, 256);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_pingpong, dim3(partner + 1), dim3(64), 0, 0, flag, ticks, clocks, failed, partner);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t = 0, c = 0;
      int f = 0;
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      hipMemcpy(&c, clocks, 8, hipMemcpyDeviceToHost);
      hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost);
      if (rep == 0) continue;  // first launch of a geometry warms up
      printf("workgroup 0 <-> workgroup %2d (%s): %s  %.1f s_memtime ticks, %.3f us (wall clock, %d kHz) per round trip of two "
             "hand-overs; kernel %.1f us for %d round trips\n",
             partner, partner % 8 == 0 ? "same XCD" : "other XCD", f ? "FAILED (spin limit)" : "ok", (double)t / REP,
             rate_khz ? (double)c / REP / (rate_khz * 1e-3) : 0.0, rate_khz, ms * 1e3, REP);
      fflush(stdout);
    }
  }
  return 0;
}
