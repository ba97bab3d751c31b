// Diagnostic only: dependent-chain latencies of the cross-lane / cross-wave primitives an FPS round is built from, one
// wave alone on a SIMD (s_memtime around 512 dependent repetitions).  Not part of the shipped library.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lat_probe.cpp -o tools/lat_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 512
__device__ unsigned long long g_out[64];
__device__ float g_sink[1024];

#define BEGIN unsigned long long t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define END(k)                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
  unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
  if (threadIdx.x == 0) g_out[k] = t1 - t0;

__global__ void k_empty(float v) { BEGIN for (int i = 0; i < REP; ++i) asm volatile("" ::: "memory"); END(0) g_sink[0] = v; }

// dependent v_add chain
__global__ void k_valu(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(v));
  END(1) g_sink[threadIdx.x] = v;
}
// v_readlane -> v_add (SGPR) chain: VALU -> SGPR -> VALU
__global__ void k_readlane(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) {
    int s;
    asm volatile("v_readlane_b32 %1, %0, 63\n\tv_add_f32 %0, %1, %0" : "+v"(v), "=s"(s));
  }
  END(2) g_sink[threadIdx.x] = v;
}
// DPP step with its two wait states
__global__ void k_dpp(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i)
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
  END(3) g_sink[threadIdx.x] = v;
}
// LDS write -> wait -> read (same address) -> wait
__global__ void k_lds(float v) {
  __shared__ float sm[1024];
  BEGIN
  for (int i = 0; i < REP; ++i) {
    asm volatile("ds_write_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"((int)(threadIdx.x * 4)) : "memory");
  }
  END(4) g_sink[threadIdx.x] = v + sm[0];
}
// LDS read only (dependent through the address)
__global__ void k_ldsread(int a) {
  __shared__ int sm[1024];
  sm[threadIdx.x] = 0;
  __syncthreads();
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");
  END(5) g_sink[threadIdx.x] = a + sm[1];
}
// v_cmp -> SALU (s_ff1) -> v_readlane with that lane select -> VALU
__global__ void k_cmp_ff1_readlane(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) {
    int s, l;
    asm volatile(
        "v_cmp_eq_f32 vcc, %0, %0\n\t"
        "s_ff1_i32_b64 %2, vcc\n\t"
        "s_nop 3\n\t"
        "v_readlane_b32 %1, %0, %2\n\t"
        "v_add_f32 %0, %1, %0"
        : "+v"(v), "=s"(s), "=s"(l) : : "vcc");
  }
  END(6) g_sink[threadIdx.x] = v;
}
// s_barrier alone (one wave: immediate) and with 4 waves arriving together
__global__ void k_barrier(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("s_barrier" ::: "memory");
  END(7) g_sink[threadIdx.x] = v;
}
// ds_bpermute broadcast of lane 5 to all
__global__ void k_bpermute(int v) {
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(20) : "memory");
  END(8) g_sink[threadIdx.x] = v;
}
// v_readfirstlane -> VALU
__global__ void k_readfirstlane(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) {
    int s;
    asm volatile("v_readfirstlane_b32 %1, %0\n\tv_add_f32 %0, %1, %0" : "+v"(v), "=s"(s));
  }
  END(9) g_sink[threadIdx.x] = v;
}
// three readlanes back to back with one consumer each (the FPS broadcast)
__global__ void k_readlane3(float v) {
  float a = v, b = v + 1, c = v + 2;
  BEGIN
  for (int i = 0; i < REP; ++i) {
    int s0, s1, s2;
    asm volatile(
        "v_readlane_b32 %3, %0, 7\n\tv_readlane_b32 %4, %1, 7\n\tv_readlane_b32 %5, %2, 7\n\t"
        "v_add_f32 %0, %3, %0\n\tv_add_f32 %1, %4, %1\n\tv_add_f32 %2, %5, %2"
        : "+v"(a), "+v"(b), "+v"(c), "=s"(s0), "=s"(s1), "=s"(s2));
  }
  END(10) g_sink[threadIdx.x] = a + b + c;
}
// v_cmp -> s_and_saveexec -> s_or exec (an exec-mask round trip around nothing)
__global__ void k_saveexec(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) {
    unsigned long long sv;
    asm volatile("v_cmp_eq_f32 vcc, %0, %0\n\ts_and_saveexec_b64 %1, vcc\n\tv_add_f32 %0, %0, %0\n\ts_or_b64 exec, exec, %1" : "+v"(v), "=s"(sv) : : "vcc", "scc");
  }
  END(11) g_sink[threadIdx.x] = v;
}
// v_pk_add chain
__global__ void k_pk(float v) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x = {v, v};
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("v_pk_add_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0\n\tv_pk_add_f32 %0, %0, %0" : "+v"(x));
  END(12) g_sink[threadIdx.x] = x[0] + x[1];
}
// independent v_add stream (4 registers)
__global__ void k_valu_indep(float v) {
  float a = v, b = v, c = v, d = v;
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  END(13) g_sink[threadIdx.x] = a + b + c + d;
}
// s_nop 1 alone
__global__ void k_nop(float v) {
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("s_nop 1\n\ts_nop 1\n\ts_nop 1\n\ts_nop 1" ::: "memory");
  END(14) g_sink[threadIdx.x] = v;
}
// SALU chain
__global__ void k_salu(int v) {
  int s = v;
  BEGIN
  for (int i = 0; i < REP; ++i) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
  END(15) g_sink[threadIdx.x] = s;
}

int main() {
  unsigned long long h[64];
  const char* names[16] = {"empty loop", "4 dependent v_add_f32", "v_readlane -> v_add (SGPR operand)", "s_nop 1 + v_max_f32_dpp",
                           "ds_write_b32 -> wait -> ds_read_b32 -> wait", "ds_read_b32 (dependent address) -> wait",
                           "v_cmp -> s_ff1 -> s_nop 3 -> v_readlane -> v_add", "s_barrier (threads arriving together)",
                           "ds_bpermute_b32 -> wait", "v_readfirstlane -> v_add", "3 x v_readlane + 3 x v_add",
                           "v_cmp -> s_and_saveexec -> v_add -> s_or exec", "4 dependent v_pk_add_f32", "4 independent v_add_f32",
                           "4 x s_nop 1", "4 dependent s_add_u32"};
  for (int threads : {64, 256, 512}) {
    printf("launch k_empty %d\n", threads); fflush(stdout);
hipLaunchKernelGGL(k_empty, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_valu %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_valu, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_readlane %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_readlane, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_dpp %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_dpp, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_lds %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_lds, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_ldsread %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_ldsread, 1, threads, 0, 0, 0);
    hipDeviceSynchronize();
    printf("launch k_cmp_ff1_readlane %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_cmp_ff1_readlane, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_barrier %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_barrier, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_bpermute %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_bpermute, 1, threads, 0, 0, 1);
    hipDeviceSynchronize();
    printf("launch k_readfirstlane %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_readfirstlane, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_readlane3 %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_readlane3, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_saveexec %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_saveexec, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_pk %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_pk, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_valu_indep %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_valu_indep, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_nop %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_nop, 1, threads, 0, 0, 1.f);
    hipDeviceSynchronize();
    printf("launch k_salu %d\n", threads); fflush(stdout);
    hipLaunchKernelGGL(k_salu, 1, threads, 0, 0, 1);
    hipDeviceSynchronize();
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h));
    printf("---- %d threads per workgroup (%d wave(s) per SIMD): shader cycles per repetition, wave 0\n", threads, (threads + 255) / 256);
    for (int k = 0; k < 16; ++k) printf("%-52s %8.1f\n", names[k], ((double)h[k] - (double)h[0]) / REP);
    fflush(stdout);
  }
  return 0;
}
