// EXPERIMENT (r03, not part of the library): see DESIGN.md "Measured dead ends".  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivision3d_amd/csrc -Itools/experiments tools/experiments/sorttest.hip -o tools/sorttest.bin
#include "devsort.h"
#include <vector>
#include <random>
#include <algorithm>
#include <cstdio>
namespace v3d { void set_last_error(const char*, ...) {} }
template <typename K>
int run(size_t n, int begin, int end, unsigned long long keymask, bool values, int seed) {
  std::mt19937_64 g(seed);
  std::vector<K> k(n); std::vector<int> v(n);
  for (size_t i = 0; i < n; ++i) { k[i] = (K)(g() & keymask); v[i] = (int)i; }
  K *dk, *dko; int *dv, *dvo; void* tmp;
  size_t tb = v3d::devsort::sort_bytes<K>(n, values);
  hipMalloc(&dk, n * sizeof(K)); hipMalloc(&dko, n * sizeof(K)); hipMalloc(&dv, n * 4); hipMalloc(&dvo, n * 4); hipMalloc(&tmp, tb);
  hipMemcpy(dk, k.data(), n * sizeof(K), hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
  int rc = v3d::devsort::sort<K>(tmp, tb, dk, dko, values ? dv : nullptr, values ? dvo : nullptr, n, begin, end, 0);
  hipDeviceSynchronize();
  std::vector<K> ok(n); std::vector<int> ov(n);
  hipMemcpy(ok.data(), dko, n * sizeof(K), hipMemcpyDeviceToHost); hipMemcpy(ov.data(), dvo, n * 4, hipMemcpyDeviceToHost);
  std::vector<int> order(n); for (size_t i = 0; i < n; ++i) order[i] = (int)i;
  K m = (end - begin >= (int)sizeof(K) * 8) ? (K)~(K)0 : (K)((((K)1 << (end - begin)) - 1) << begin);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return (k[a] & m) < (k[b] & m); });
  int bad = rc != 0;
  for (size_t i = 0; i < n && !bad; ++i) { if (ok[i] != k[order[i]]) bad = 1; if (values && ov[i] != order[i]) bad = 2; }
  // timing
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 20; ++r) v3d::devsort::sort<K>(tmp, tb, dk, dko, values ? dv : nullptr, values ? dvo : nullptr, n, begin, end, 0);
  hipEventRecord(e1, 0); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("K=%zu n=%zu bits[%d,%d) mask=%llx values=%d rc=%d %s  %.1f us\n", sizeof(K), n, begin, end, keymask, (int)values, rc, bad ? "MISMATCH" : "ok", ms / 20 * 1e3);
  hipFree(dk); hipFree(dko); hipFree(dv); hipFree(dvo); hipFree(tmp);
  return bad;
}
int scan_test(size_t n) {
  std::vector<int> a(n); for (size_t i = 0; i < n; ++i) a[i] = (int)(i * 7919 % 5);
  int* d; void* tmp; size_t tb = v3d::devsort::scan_bytes(n);
  hipMalloc(&d, n * 4); hipMalloc(&tmp, tb); hipMemcpy(d, a.data(), n * 4, hipMemcpyHostToDevice);
  int rc = v3d::devsort::exclusive_scan(tmp, tb, d, d, 3, n, 0); hipDeviceSynchronize();
  std::vector<int> o(n); hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
  long long run = 3; int bad = rc != 0;
  for (size_t i = 0; i < n && !bad; ++i) { if (o[i] != (int)run) bad = 1; run += a[i]; }
  printf("scan n=%zu rc=%d %s\n", n, rc, bad ? "MISMATCH" : "ok"); hipFree(d); hipFree(tmp); return bad;
}
int main() {
  int bad = 0;
  bad |= run<unsigned long long>(1000, 0, 64, ~0ull, true, 1);
  bad |= run<unsigned long long>(480000, 0, 64, 0x1ffffffull, true, 2);
  bad |= run<unsigned long long>(480000, 0, 64, 0x3c07f01fc007full, true, 3);
  bad |= run<unsigned long long>(480000, 0, 64, ~0ull, true, 4);
  bad |= run<unsigned long long>(5000, 0, 64, 0xff00ull, false, 5);
  bad |= run<unsigned long long>(1, 0, 64, ~0ull, true, 6);
  bad |= run<unsigned long long>(70000, 3, 29, ~0ull, true, 7);
  bad |= run<unsigned long long>(4097, 0, 64, 0ull, true, 8);
  bad |= run<unsigned>(18240000, 0, 19, 0x7ffffu, true, 9);
  bad |= run<unsigned>(300000, 0, 32, ~0u, true, 10);
  bad |= run<unsigned long long>(3000000, 0, 44, 0xfffffffffffull, true, 11);
  for (size_t n : {1ul, 100ul, 4096ul, 4097ul, 480000ul, 18240000ul}) bad |= scan_test(n);
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad;
}
