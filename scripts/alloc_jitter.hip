// alloc_jitter.hip -- how long do the driver-side calls of a first solver call take on this box, and do they have slow WINDOWS?
// (the driver's run recorded a 441 ms first call of config 5 where the builder's runs had 80; round 5 reproduced it once
// in six default runs -- orientation 88 instead of 7.5 ms, tables 227 instead of 28, every kernel time unchanged.)
// Loops for `seconds`: hipMalloc + hipFree of `mb` MB, a 4-byte device-to-host copy after an empty kernel, an empty kernel + stream sync;
// prints the percentiles of each and every sample beyond 10 x its median with its time stamp.
// build: hipcc --offload-arch=gfx950 -O2 scripts/alloc_jitter.hip -o scripts/tmp/alloc_jitter ; run: alloc_jitter [seconds=20] [mb=1024]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void nop_kernel(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) *p = 1; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
  const size_t mb = argc > 2 ? (size_t)atoll(argv[2]) : 1024;
  int *d = nullptr, h = 0;
  if (hipMalloc(&d, 4) != hipSuccess) return 1;
  struct S { double t, v; };
  std::vector<S> a, f, c, k;
  const double t0 = now_ms();
  while (now_ms() - t0 < seconds * 1e3) {
    void *p = nullptr;
    double t = now_ms();
    if (hipMalloc(&p, mb << 20) != hipSuccess) return 2;
    a.push_back({t - t0, now_ms() - t});
    t = now_ms();
    (void)hipFree(p);
    f.push_back({t - t0, now_ms() - t});
    t = now_ms();
    hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    c.push_back({t - t0, now_ms() - t});
    t = now_ms();
    hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, 0, (int *)nullptr);
    (void)hipStreamSynchronize(0);
    k.push_back({t - t0, now_ms() - t});
  }
  auto report = [&](const char *name, std::vector<S> &x) {
    std::vector<double> v;
    for (auto &s : x) v.push_back(s.v);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    printf("%-28s n %6zu  median %8.3f ms  p90 %8.3f  p99 %8.3f  max %8.3f\n", name, v.size(), med, v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
    int shown = 0;
    for (auto &s : x)
      if (s.v > 10 * med && shown++ < 12) printf("    at %9.1f ms: %8.3f ms\n", s.t, s.v);
  };
  char name[64];
  snprintf(name, sizeof name, "hipMalloc(%zu MB)", mb);
  report(name, a);
  report("hipFree", f);
  report("kernel + 4-byte D2H copy", c);
  report("kernel + stream sync", k);
  return 0;
}
