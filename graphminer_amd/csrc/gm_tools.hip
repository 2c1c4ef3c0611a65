// gm_tools.hip -- tooling behind the C ABI: the set-operation batch (parity tests of the wave64 primitives), the R-MAT key generator,
// the PMC / stream-ceiling / issue-rate calibration kernels, the wave-primitive self test.
#include "gm_host.h"
#include "gm_setops.h"
#include "gm_flat.h"
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

// Kernel constants the byte model of bench.py (own_bytes_device) depends on, by name -- read from the headers the kernels are compiled
// with, so the model cannot drift from them.
extern "C" int gm_constant(const char *name, int64_t *value) {
  if (!name || !value) return GM_ERR_INVALID;
  static const struct { const char *name; long long v; } tab[] = {
      {"tct_stage_max", kTctStageMax},       {"topo_min_mean_row", kTopoMinMeanRow}, {"motif_trim_min_list", kMotifTrimMinList},
      {"cb_min_deg", kCbMinDeg},             {"cb_max_deg", kCbMaxDeg},               {"long_list", kLongList},
      {"stage_cap", kStageCap},              {"default_chunk", kDefaultChunk},        {"mma_words_small", kMmaWordsS},
      {"mma_words_big", kMmaWordsL},         {"core_h_default", kCoreHDefault},  {"wide_min_words", kWideMinWordsDefault},
              {"wide_max_deg", kWideMaxDeg},           {"bit_words", kBitWords},
  };
  for (const auto &e : tab)
    if (strcmp(e.name, name) == 0) {
      *value = e.v;
      return GM_OK;
    }
  g_last_error = std::string("gm_constant: unknown name '") + name + "'";
  return GM_ERR_INVALID;
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
  HIP_TRY(dev_malloc(&d, sizeof(int) * 512));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[512];
  hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  dev_free(d);
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

// ------------------------------------------------------------------------------------------------
// Issue-rate calibration (tooling; VERDICT r2: "calibrate issue costs the way FETCH_SIZE was calibrated").
// W waves per SIMD on every CU each execute `iters` x 64 instructions of ONE kind over 8 independent registers (inline asm, so the
// compiler neither folds nor reorders them) between two s_memtime reads.  From the per-wave shader cycles c (median over the waves):
//   cycles per wave-instruction            c / (64 iters)          -- the issue interval one wave sees
//   wave-instructions per cycle and SIMD   W * 64 iters / c        -- the unit's throughput (VALU, LDS: per SIMD; SALU: x4 = per CU)
// Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` the same launches tell what
// the SQ counters report for a unit that is KNOWN to be saturated (scripts/issue_calibration.py -> profiles/r03/issue_calibration.txt).
// The loop itself adds 3 scalar instructions (s_add, s_cmp, s_cbranch) per 64 measured ones.
enum : int {
  CAL_V_ADD = 0, CAL_V_MUL_LO = 1, CAL_V_MUL_U24 = 2, CAL_V_CMP_SGPR = 3, CAL_V_DPP = 4, CAL_S_ADD = 5, CAL_DS_READ_B128 = 6,
  CAL_DS_READ_B32 = 7, CAL_DS_READ_B32_CONFLICT = 8, CAL_V_S_MIX = 9, CAL_V_READLANE = 10, CAL_V_MBCNT = 11, CAL_V_BCNT = 12,
  CAL_V_CMP_SDWA = 13, CAL_DS_WRITE_B32 = 14, CAL_V_CNDMASK = 15, CAL_V_CNDMASK_SGPR = 16, CAL_V_AND = 17, CAL_V_ASHR = 18, CAL_V_MIN = 19,
  CAL_V_MAD_U24 = 20, CAL_V_ADD3 = 21, CAL_V_CMP_VCC = 22, CAL_V_CMP_CNDMASK = 23, CAL_V_MOV = 24, CAL_V_SUB_ASHR_AND_ADD = 25, CAL_V_XOR_SGPR = 26,
  CAL_S_BCNT1 = 27, CAL_KINDS = 28
};

#define GM_CAL8(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define GM_CAL64(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void issue_calib_kernel(int iters, unsigned long long *__restrict__ cycles, unsigned *__restrict__ sink) {
  extern __shared__ unsigned dyn_lds[];  // sized by the host so that exactly W workgroups fit a CU
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (unsigned)(tid * 8 + i) | 1u;
  unsigned s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = (unsigned)(blockIdx.x + i);
  unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned k = (unsigned)iters | 3u;
  // LDS addresses: conflict-free (consecutive 16 B / 4 B per lane) or one bank for 32 lanes (stride 256 B: 2 x 32-way)
  for (int i = tid; i < 4096; i += 256) dyn_lds[i] = (unsigned)i;
  __syncthreads();
  const unsigned a128 = (unsigned)tid * 16u, a32 = (unsigned)tid * 4u, aconf = ((unsigned)lane * 256u) & 16383u;
  uint4 w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = make_uint4(0, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == CAL_V_ADD) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MUL_LO) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MUL_U24) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_CMP_SGPR) {
#define OP(i) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(m[i]) : "v"(v[i]), "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_DPP) {
#define OP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_S_ADD) {
#define OP(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(k) : "scc");
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_DS_READ_B128) {
#define OP(i) asm volatile("ds_read_b128 %0, %1" : "=v"(w[i]) : "v"(a128));
      GM_CAL64(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == CAL_DS_READ_B32) {
#define OP(i) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(a32));
      GM_CAL64(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == CAL_DS_READ_B32_CONFLICT) {
#define OP(i) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(aconf));
      GM_CAL64(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == CAL_V_S_MIX) {  // 32 VALU + 32 SALU interleaved: do they issue side by side?
#define OP(i) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(v[i]), "+s"(s[i]) : "v"(k), "s"(k) : "scc");
      GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_READLANE) {
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 17" : "=s"(s[i]) : "v"(v[i]));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MBCNT) {
#define OP(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(v[i]) : "s"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_BCNT) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_CMP_SDWA) {
#define OP(i) asm volatile("v_cmp_eq_u16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0" : "=s"(m[i]) : "v"(v[i]), "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_DS_WRITE_B32) {
#define OP(i) asm volatile("ds_write_b32 %0, %1" : : "v"(a32), "v"(v[i]) : "memory");
      GM_CAL64(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == CAL_V_CNDMASK) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(k) : "vcc");
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_CNDMASK_SGPR) {  // the mask in an SGPR pair (VOP3 form), as a v_cmp ... -> s[a:b] leaves it
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(k), "s"(m[0]));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_AND) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_ASHR) {
#define OP(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(v[i]));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MIN) {
#define OP(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MAD_U24) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_ADD3) {
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_CMP_VCC) {
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(k) : "vcc");
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_CMP_CNDMASK) {  // the select idiom of a branch-free bisection step: 32 compares + 32 selects
#define OP(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(k) : "vcc");
      GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP) GM_CAL8(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_MOV) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(k));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_SUB_ASHR_AND_ADD) {  // the same select as arithmetic: d = x - k; mask = d >> 31; x += mask & k  (16 selects = 64 instructions)
#define OP(i) asm volatile("v_sub_u32 %1, %0, %2\n\tv_ashrrev_i32 %1, 31, %1\n\tv_and_b32 %1, %1, %2\n\tv_add_u32 %0, %0, %1" : "+v"(v[i]), "+v"(w[i].x) : "v"(k));
      GM_CAL8(OP) GM_CAL8(OP)
#undef OP
    } else if constexpr (KIND == CAL_V_XOR_SGPR) {  // a VALU with a scalar source operand
#define OP(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[i]) : "s"(s[i]));
      GM_CAL64(OP)
#undef OP
    } else if constexpr (KIND == CAL_S_BCNT1) {
#define OP(i) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s[i]) : "s"(m[i]) : "scc");
      GM_CAL64(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= v[i] ^ s[i] ^ (unsigned)m[i] ^ (unsigned)(m[i] >> 32) ^ w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
  if (acc == 0x12345u) sink[0] = acc;  // (keeps every chain alive)
  if (lane == 0) {
    // per wave: start, end (s_memtime) and where it ran: HW_ID (wave / simd / cu / sh / se fields) and the XCC id
    unsigned hwid = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long *o = cycles + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 3;
    o[0] = t0;
    o[1] = t1;
    o[2] = ((unsigned long long)xcc << 32) | hwid;
  }
}

// residency: the largest number of waves of the launch that were in flight on one SIMD at the same time (from the per-wave start /
// end stamps and HW_ID): what the "W waves per SIMD" of the request really was
extern "C" int gm_issue_calib(int kind, int waves_per_simd, int iters, double *cycles_per_wave_inst, double *inst_per_cycle_simd, double *ms,
                              double *residency) {
  if (kind < 0 || kind >= CAL_KINDS || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !cycles_per_wave_inst || !inst_per_cycle_simd)
    return GM_ERR_INVALID;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int W = waves_per_simd, grid = cus * W;
  // exactly W workgroups of 4 waves (one per SIMD) per CU: each claims 1/W of the 160 KB of LDS (at least the 16 KB the kernel touches)
  size_t lds = std::max<size_t>(16384, ((size_t)(160 * 1024) / (size_t)W) & ~(size_t)1023);
  if (const char *e = gm_sweep_env("GM_CAL_LDS")) lds = std::max<size_t>(16384, (size_t)atoll(e));  // (diagnostics: let the dispatcher pack as it likes)
  DevBuf<unsigned long long> cyc;
  DevBuf<unsigned> sink;
  HIP_TRY(cyc.alloc((size_t)grid * 4 * 3));
  HIP_TRY(sink.alloc(4));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float best_ms = 0.f;
  std::vector<unsigned long long> h((size_t)grid * 4 * 3);
  for (int rep = 0; rep < 3; ++rep) {  // the last repetition is the one reported (the first warms the instruction cache / clocks)
    HIP_TRY(hipEventRecord(e0, 0));
#define GM_CAL_CASE(K) case K: \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&issue_calib_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((issue_calib_kernel<K>), dim3((unsigned)grid), dim3(256), lds, 0, iters, cyc.p, sink.p); break;
    switch (kind) {
      GM_CAL_CASE(0) GM_CAL_CASE(1) GM_CAL_CASE(2) GM_CAL_CASE(3) GM_CAL_CASE(4) GM_CAL_CASE(5) GM_CAL_CASE(6) GM_CAL_CASE(7)
      GM_CAL_CASE(8) GM_CAL_CASE(9) GM_CAL_CASE(10) GM_CAL_CASE(11) GM_CAL_CASE(12) GM_CAL_CASE(13) GM_CAL_CASE(14) GM_CAL_CASE(15)
      GM_CAL_CASE(16) GM_CAL_CASE(17) GM_CAL_CASE(18) GM_CAL_CASE(19) GM_CAL_CASE(20) GM_CAL_CASE(21) GM_CAL_CASE(22) GM_CAL_CASE(23)
      GM_CAL_CASE(24) GM_CAL_CASE(25) GM_CAL_CASE(26) GM_CAL_CASE(27)
      default: break;
    }
#undef GM_CAL_CASE
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(&best_ms, e0, e1));
  }
  HIP_TRY(hipMemcpy(h.data(), cyc.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const size_t nw = (size_t)grid * 4;
  std::vector<unsigned long long> dur(nw);
  for (size_t i = 0; i < nw; ++i) dur[i] = h[3 * i + 1] - h[3 * i];
  std::sort(dur.begin(), dur.end());
  const double med = (double)dur[nw / 2], n = 64.0 * (double)iters;
  // waves in flight per SIMD: sweep over the (start, +1) / (end, -1) events of the waves that share (xcc, se, sh, cu, simd)
  // (HW_ID, gfx9 layout: wave_id 3:0, simd_id 5:4, pipe_id 7:6, cu_id 11:8, sh_id 12, se_id 15:13; s_memtime is per XCC)
  double res_sum = 0.0;
  size_t res_n = 0;
  {
    std::vector<std::pair<unsigned long long, std::pair<unsigned long long, int>>> ev;  // (simd key, (time, +-1))
    ev.reserve(2 * nw);
    for (size_t i = 0; i < nw; ++i) {
      const unsigned long long id = h[3 * i + 2];
      const unsigned long long key = ((id >> 32) << 16) | ((id & 0xffffull) >> 4);  // xcc | se, sh, cu, pipe, simd
      ev.push_back({key, {h[3 * i], +1}});
      ev.push_back({key, {h[3 * i + 1], -1}});
    }
    std::sort(ev.begin(), ev.end(), [](const auto &x, const auto &y) {
      if (x.first != y.first) return x.first < y.first;
      if (x.second.first != y.second.first) return x.second.first < y.second.first;
      return x.second.second < y.second.second;
    });
    for (size_t i = 0; i < ev.size();) {
      size_t j = i;
      int cur = 0, mx = 0;
      for (; j < ev.size() && ev[j].first == ev[i].first; ++j) {
        cur += ev[j].second.second;
        mx = std::max(mx, cur);
      }
      res_sum += mx;
      ++res_n;
      i = j;
    }
  }
  const double res = res_n ? res_sum / (double)res_n : 0.0;
  *cycles_per_wave_inst = med / n;
  *inst_per_cycle_simd = (res > 0 ? res : (double)W) * n / med;  // (with the MEASURED residency, not the requested one)
  if (ms) *ms = best_ms;
  if (residency) *residency = res;
  return GM_OK;
}

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_tools_kernel() {}
void gm_touch_tools() { hipLaunchKernelGGL(gm_touch_tools_kernel, dim3(1), dim3(1), 0, 0); }
