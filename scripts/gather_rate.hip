// Random-address load rates on gfx950 by table size and element width: what bounds every per-entry gather of the setup passes
// (orientation: the degree of an entry's target; edge descriptors: its offsets; renumbering: its new id -- all ~100 G entries/s on
// R-MAT-24, profiles/r04/ab_setup_orient_relabel.txt).  Each thread sums PER independent random elements (eight loads in flight).
//   hipcc --offload-arch=gfx950 -O3 scripts/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (unsigned)x;
}

template <class T>
__global__ __launch_bounds__(256) void gather_kernel(const T *__restrict__ tab, unsigned long long n /* power of two */, int rounds, unsigned long long *sink) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  for (int r = 0; r < rounds; ++r) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tab[mix(gid * 0x9E3779B97F4A7C15ull + (unsigned long long)(r * 8 + k)) & (n - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (unsigned long long)v[k];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}

template <class T>
static void sweep(const char *name) {
  unsigned long long *sink;
  CK(hipMalloc(&sink, 8));
  printf("%s elements\n", name);
  for (unsigned long long bytes : {1ull << 18, 1ull << 20, 1ull << 22, 1ull << 24, 1ull << 26, 1ull << 28}) {
    const unsigned long long n = bytes / sizeof(T);
    T *tab;
    CK(hipMalloc(&tab, bytes));
    CK(hipMemset(tab, 1, bytes));
    const int blocks = 256 * 16, rounds = 32;
    const double total = (double)blocks * 256 * rounds * 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<T>), dim3(blocks), dim3(256), 0, 0, tab, n, 2, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<T>), dim3(blocks), dim3(256), 0, 0, tab, n, rounds, sink);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("  table %7.2f MB   %8.3f ms   %7.1f G loads/s\n", bytes / 1048576.0, ms, total / (ms * 1e6));
    CK(hipFree(tab));
  }
  CK(hipFree(sink));
}

int main() {
  sweep<unsigned char>("1-byte");
  sweep<unsigned>("4-byte");
  sweep<unsigned long long>("8-byte");
  return 0;
}
