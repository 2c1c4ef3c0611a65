// gm_tools.hip -- tooling behind the C ABI: the set-operation batch (parity tests of the wave64 primitives), the R-MAT key generator,
// the PMC / stream-ceiling / issue-rate calibration kernels, the wave-primitive self test.
#include "gm_host.h"
#include "gm_setops.h"
using namespace gm;

// ------------------------------------------------------------------------------------------------
// set-op batch (one wave per pair)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void setop_kernel(int op, long long npairs, const int *__restrict__ vals,
                                                    const long long *__restrict__ ab, const long long *__restrict__ ae,
                                                    const long long *__restrict__ bb, const long long *__restrict__ be,
                                                    const int *__restrict__ upper, const int *__restrict__ skip,
                                                    unsigned *__restrict__ out_num, int *__restrict__ out_vals) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long i = wave; i < npairs; i += nwaves) {
    const int *A = vals + ab[i];
    const int *B = vals + bb[i];
    const int a = (int)(ae[i] - ab[i]), b = (int)(be[i] - bb[i]);
    const int up = upper ? upper[i] : 0x7fffffff;
    const int sk = skip ? skip[i] : -1;
    int *O = out_vals ? out_vals + ab[i] : nullptr;
    unsigned r = 0;
    switch (op) {
      case GM_OP_INTERSECT_NUM: r = (unsigned)wave_sum((int)wave_intersect_num(A, a, B, b)); break;
      case GM_OP_INTERSECT_NUM_UPPER: r = (unsigned)wave_sum((int)wave_intersect_num_upper(A, a, B, b, up)); break;
      case GM_OP_INTERSECT_SET: r = (unsigned)wave_intersect_set(A, a, B, b, O); break;
      case GM_OP_INTERSECT_SET_UPPER: r = (unsigned)wave_intersect_set_upper(A, a, B, b, up, O); break;
      case GM_OP_DIFFERENCE_NUM: r = (unsigned)wave_sum((int)wave_difference_num(A, a, B, b, sk)); break;
      case GM_OP_DIFFERENCE_NUM_UPPER: r = (unsigned)wave_sum((int)wave_difference_num_upper(A, a, B, b, sk, up)); break;
      case GM_OP_DIFFERENCE_SET: r = (unsigned)wave_difference_set(A, a, B, b, sk, O); break;
      case GM_OP_DIFFERENCE_SET_UPPER: r = (unsigned)wave_difference_set_upper(A, a, B, b, sk, up, O); break;
      case GM_OP_COUNT_SMALLER: r = (unsigned)wave_sum((int)wave_count_smaller(up, A, a)); break;
      default: break;
    }
    if (lane == 0) out_num[i] = r;
  }
}

extern "C" int gm_setop_batch(int op, int64_t npairs, const int32_t *d_values, const int64_t *d_a_begin, const int64_t *d_a_end,
                              const int64_t *d_b_begin, const int64_t *d_b_end, const int32_t *d_upper, const int32_t *d_skip,
                              uint32_t *d_out_num, int32_t *d_out_values, void *stream) {
  if (op < 0 || op > GM_OP_COUNT_SMALLER || npairs < 0) return GM_ERR_INVALID;
  if (npairs == 0) return GM_OK;
  if (!d_values || !d_a_begin || !d_a_end || !d_b_begin || !d_b_end || !d_out_num) return GM_ERR_INVALID;
  const bool is_set = (op == GM_OP_INTERSECT_SET || op == GM_OP_INTERSECT_SET_UPPER || op == GM_OP_DIFFERENCE_SET ||
                       op == GM_OP_DIFFERENCE_SET_UPPER);
  if (is_set && !d_out_values) return GM_ERR_INVALID;
  const long long blocks = std::min<long long>((npairs + 3) / 4, 8192);
  hipLaunchKernelGGL(setop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, op, (long long)npairs, d_values,
                     (const long long *)d_a_begin, (const long long *)d_a_end, (const long long *)d_b_begin,
                     (const long long *)d_b_end, d_upper, d_skip, d_out_num, d_out_values);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// R-MAT key generator (tooling)
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long gm_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void rmat_kernel(int scale, long long n_edges, unsigned long long seed,
                                                   unsigned long long *__restrict__ keys) {
  const unsigned TA = 2448131358u;  // floor(0.57 * 2^32)
  const unsigned TB = 3264175144u;  // TA + floor(0.19 * 2^32)
  const unsigned TC = 4080218930u;  // TB + floor(0.19 * 2^32)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_edges; i += stride) {
    const unsigned long long h = gm_mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
    unsigned long long s = 0, d = 0;
    for (int l = 0; l < scale; ++l) {
      const unsigned r = (unsigned)(gm_mix64(h + 0xD1B54A32D192ED03ull * (unsigned long long)(l + 1)) >> 32);
      const unsigned q = (r < TA) ? 0u : (r < TB) ? 1u : (r < TC) ? 2u : 3u;
      s = (s << 1) | (q >> 1);
      d = (d << 1) | (q & 1u);
    }
    if (s == d) {
      keys[2 * i] = ~0ull;
      keys[2 * i + 1] = ~0ull;
    } else {
      keys[2 * i] = (s << 32) | d;
      keys[2 * i + 1] = (d << 32) | s;
    }
  }
}

extern "C" int gm_rmat_keys(int scale, int64_t n_edges, uint64_t seed, uint64_t *d_keys, void *stream) {
  if (scale < 1 || scale > 30 || n_edges < 0 || !d_keys) return GM_ERR_INVALID;
  if (n_edges == 0) return GM_OK;
  const long long blocks = std::min<long long>((n_edges + 255) / 256, 65536);
  hipLaunchKernelGGL(rmat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scale, (long long)n_edges,
                     (unsigned long long)seed, (unsigned long long *)d_keys);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// PMC calibration stream: every lane reads one dword per iteration (the access width of the mining kernels'
// key loads); n*4 bytes are read exactly once, so FETCH_SIZE / (4n) gives the counter's scale for this width.
__global__ __launch_bounds__(256) void calib_stream_kernel(const int *__restrict__ buf, long long n, unsigned long long *out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += (unsigned)buf[i];
  acc = wave_sum_u64(acc);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

extern "C" int gm_calib_stream(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream) {
  if (!d_buf || !d_out || n < 0) return GM_ERR_INVALID;
  hipLaunchKernelGGL(calib_stream_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, d_buf, (long long)n,
                     (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// Stream ceiling: the fastest plain read this library can issue (16 B per lane, grid-stride, 8 workgroups per CU), used by
// bench.py as the MEASURED HBM ceiling next to the 8 TB/s spec (SURVEY.md 8d "Bounding roofline").
typedef int gm_v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_ceiling_kernel(const gm_v4i *__restrict__ buf, long long n4, unsigned long long *out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  gm_v4i acc = {0, 0, 0, 0};
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    const gm_v4i a = __builtin_nontemporal_load(buf + i), b = __builtin_nontemporal_load(buf + i + stride);
    const gm_v4i c = __builtin_nontemporal_load(buf + i + 2 * stride), d = __builtin_nontemporal_load(buf + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n4; i += stride) acc ^= buf[i];
  const unsigned long long s = wave_sum_u64((unsigned long long)(unsigned)(acc.x ^ acc.y ^ acc.z ^ acc.w));
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

extern "C" int gm_stream_ceiling(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream) {
  if (!d_buf || !d_out || n < 0 || ((uintptr_t)d_buf & 15)) return GM_ERR_INVALID;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  hipLaunchKernelGGL(stream_ceiling_kernel, dim3((unsigned)(cus * 8)), dim3(256), 0, (hipStream_t)stream, (const gm_v4i *)d_buf,
                     (long long)(n / 4), (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// wave-primitive self test
// ------------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int *out) {
  __shared__ int lds[64];
  const int lane = threadIdx.x;
  const int x = (lane * 7 + 3) % 11;
  out[lane] = wave_incl_scan_add(x);
  const int y = (lane % 9 == 0) ? lane + 1 : 0;
  out[64 + lane] = wave_incl_scan_max(y);
  const unsigned long long m = __ballot((lane % 3) == 1);
  out[128 + lane] = rank_below(m);
  out[192 + lane] = lane_id();
  lds[lane] = lane * 3;  // ascending
  wave_sync();
  int pos;
  const bool f = contains(&lds[0], 64, lane * 2, &pos);
  out[256 + lane] = f ? pos : -1;
  out[320 + lane] = lower_bound(&lds[0], 64, lane * 2);
  out[384 + lane] = (int)wave_sum_u64((unsigned long long)lane + (1ull << 33));  // low word of 64*2^33 + 2016
  out[448 + lane] = (int)(wave_sum_u64((unsigned long long)lane + (1ull << 33)) >> 32);
}

extern "C" int gm_selftest(int device, int *n_fail) {
  if (n_fail) *n_fail = -1;
  HIP_TRY(hipSetDevice(device));
  int *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(int) * 512));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[512];
  hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "selftest", __FILE__, __LINE__);
  int bad = 0, acc = 0, mx = 0, rk = 0;
  for (int l = 0; l < 64; ++l) {
    acc += (l * 7 + 3) % 11;
    bad += h[l] != acc;
    const int y = (l % 9 == 0) ? l + 1 : 0;
    mx = std::max(mx, y);
    bad += h[64 + l] != mx;
    bad += h[128 + l] != rk;
    rk += (l % 3) == 1;
    bad += h[192 + l] != l;
    const int key = l * 2;
    const int expect_pos = (key % 3 == 0 && key / 3 < 64) ? key / 3 : -1;
    bad += h[256 + l] != expect_pos;
    int lb = 0;
    while (lb < 64 && lb * 3 < key) ++lb;
    bad += h[320 + l] != lb;
    const unsigned long long tot = 64ull * (1ull << 33) + 2016ull;
    bad += h[384 + l] != (int)(unsigned)tot;
    bad += h[448 + l] != (int)(tot >> 32);
  }
  if (n_fail) *n_fail = bad;
  if (bad) { g_last_error = "wave primitive self test mismatch"; return GM_ERR_HIP; }
  return GM_OK;
}
