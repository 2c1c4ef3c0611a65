// Random-address atomic rates on gfx950: device (agent) scope against workgroup scope on a counter copy private to the XCD.
// The key stream's count / place passes (gm_tables.hip, kst_rows_kernel) spend 3/4 of their time in one random global atomic per
// in-edge task (profiles/r04/ab_setup_orient_relabel.txt); this measures what a per-XCD copy with L2-scope atomics would buy, and
// checks that such atomics from different workgroups of one XCD add up (sum over the copies == atomics issued).
//   hipcc --offload-arch=gfx950 -O3 scripts/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (unsigned)x;
}
__device__ __forceinline__ int xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

// MODE 0: agent scope, one table; 1: agent scope, the XCD's copy; 2: workgroup scope, the XCD's copy.  RET: the value is used.
template <class T, int MODE, bool RET>
__global__ __launch_bounds__(256) void rate_kernel(T *tab, unsigned long long n /* counters per copy, power of two */, int per_thread, unsigned long long *sink) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  T *t = MODE == 0 ? tab : tab + (unsigned long long)xcc_id() * n;
  T acc = 0;
  for (int k = 0; k < per_thread; ++k) {
    const unsigned long long i = mix(gid * 0x9E3779B97F4A7C15ull + (unsigned long long)k) & (n - 1);
    if (MODE == 2) {
      const T o = __hip_atomic_fetch_add(&t[i], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (RET) acc += o;
    } else {
      const T o = __hip_atomic_fetch_add(&t[i], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (RET) acc += o;
    }
  }
  if (RET && acc == (T)0x7fffffff) *sink = (unsigned long long)acc;
}
template <class T>
__global__ void sum_kernel(const T *tab, unsigned long long n, unsigned long long *out) {
  unsigned long long s = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) s += tab[i];
  atomicAdd(out, s);
}

template <class T, int MODE, bool RET>
static void run(const char *what, T *tab, unsigned long long n, unsigned long long *d_out) {
  const int blocks = 256 * 8, per_thread = 128;
  const unsigned long long total = (unsigned long long)blocks * 256 * per_thread;
  CK(hipMemset(tab, 0, sizeof(T) * n * 8));
  CK(hipMemset(d_out, 0, 16));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((rate_kernel<T, MODE, RET>), dim3(blocks), dim3(256), 0, 0, tab, n, 8, d_out + 1);  // warm
  CK(hipDeviceSynchronize());
  CK(hipMemset(tab, 0, sizeof(T) * n * 8));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((rate_kernel<T, MODE, RET>), dim3(blocks), dim3(256), 0, 0, tab, n, per_thread, d_out + 1);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  hipLaunchKernelGGL((sum_kernel<T>), dim3(1024), dim3(256), 0, 0, tab, n * 8, d_out);
  unsigned long long got = 0;
  CK(hipMemcpy(&got, d_out, 8, hipMemcpyDeviceToHost));
  printf("  %-44s %8.3f ms  %7.1f G atomics/s   sum %s\n", what, ms, total / (ms * 1e6), got == total ? "ok" : "WRONG");
  if (got != total) printf("     issued %llu, counted %llu\n", total, got);
}

template <class T>
static void sweep(const char *name) {
  unsigned long long *d_out;
  CK(hipMalloc(&d_out, 16));
  for (unsigned long long n : {1ull << 16, 1ull << 19, 1ull << 22, 1ull << 24}) {
    T *tab;
    CK(hipMalloc(&tab, sizeof(T) * n * 8));
    printf("%s counters, %llu per copy (%.1f MB per copy)\n", name, n, sizeof(T) * n / 1048576.0);
    run<T, 0, false>("agent scope, one table", tab, n, d_out);
    run<T, 1, false>("agent scope, the XCD's copy", tab, n, d_out);
    run<T, 2, false>("workgroup scope, the XCD's copy", tab, n, d_out);
    run<T, 0, true>("agent scope, one table, returning", tab, n, d_out);
    run<T, 2, true>("workgroup scope, the XCD's copy, returning", tab, n, d_out);
    CK(hipFree(tab));
  }
  CK(hipFree(d_out));
}

int main() {
  sweep<unsigned>("32-bit");
  sweep<unsigned long long>("64-bit");
  return 0;
}
