// gm_wave.h -- wavefront-64 primitives for gfx950 (CDNA4), hand-written.
//
// These replace the reference's 32-lane warp helpers, re-derived for wave64 rather
// than translated:
//   __ballot_sync + __popc(mask << (31-lane)) prefix rank  (include/set_intersect.cuh:63-67)
//       -> 64-bit ballot + v_mbcnt_lo/hi                     (gm_rank_below)
//   warp_reduce via __shfl_down_sync                        (include/operations.cuh:7-16)
//       -> DPP row_shr / row_bcast scans + v_readlane       (gm_wave_sum*)
//   cub::BlockReduce + atomicAdd                            (src/triangle/gpu_kernels/bs_warp_edge.cuh:16-17)
//       -> per-wave DPP reduce + one global atomic per wave
//   __syncwarp                                              (include/set_intersect.cuh:56)
//       -> wave lock-step + a wavefront-scope fence (compiler ordering only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GM_WAVE 64

namespace gm {

// All 64 lanes execute DS instructions of one wave in program order, so intra-wave
// LDS hand-offs need no s_barrier; this only stops the compiler from reordering.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` strictly below this lane (64-bit ballot masks)
__device__ __forceinline__ int rank_below(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int readlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ __forceinline__ int readfirst(int x) { return __builtin_amdgcn_readfirstlane(x); }

// DPP controls (gfx9 encoding): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
#define GM_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)

// inclusive prefix sum across the 64 lanes (all lanes must be active)
__device__ __forceinline__ int wave_incl_scan_add(int x) {
  x += GM_DPP(0, x, 0x111, 0xf);
  x += GM_DPP(0, x, 0x112, 0xf);
  x += GM_DPP(0, x, 0x114, 0xf);
  x += GM_DPP(0, x, 0x118, 0xf);
  x += GM_DPP(0, x, 0x142, 0xa);  // lane 15 of rows 0,2 -> rows 1,3
  x += GM_DPP(0, x, 0x143, 0xc);  // lane 31 -> rows 2,3
  return x;
}

// inclusive prefix max across the 64 lanes for NON-NEGATIVE values (identity 0)
__device__ __forceinline__ int wave_incl_scan_max(int x) {
  x = max(x, GM_DPP(0, x, 0x111, 0xf));
  x = max(x, GM_DPP(0, x, 0x112, 0xf));
  x = max(x, GM_DPP(0, x, 0x114, 0xf));
  x = max(x, GM_DPP(0, x, 0x118, 0xf));
  x = max(x, GM_DPP(0, x, 0x142, 0xa));
  x = max(x, GM_DPP(0, x, 0x143, 0xc));
  return x;
}

// wave-wide sum, result valid in every lane
__device__ __forceinline__ int wave_sum(int x) { return readlane(wave_incl_scan_add(x), 63); }
__device__ __forceinline__ int wave_max_nonneg(int x) { return readlane(wave_incl_scan_max(x), 63); }

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  // 64-bit add = two 32-bit DPP scans with a carry fix-up would be fiddly; the
  // reduction runs once per wave at kernel end, so use the xor-butterfly on both halves.
  for (int o = 32; o >= 1; o >>= 1) {
    unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o, 64);
    unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o, 64);
    v += ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

}  // namespace gm
