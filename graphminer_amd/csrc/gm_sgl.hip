// gm_sgl.hip -- nested SgL patterns (diamond listing form / rectangle / house / pentagon) on the wave64 set-op primitives.
//
// First correct MI355X versions of the "next" SgL rows. Loop nests and symmetry breaking
// follow the CPU oracle semantics:
//   rectangle  src/sgl/cpu_kernels/rectangle.h:1-11   (GPU shape: src/sgl/gpu_kernels/rectangle_nested.cuh:3)
//   house      src/sgl/cpu_kernels/house.h:1-16        (house_edge_warp_nested.cuh:4)
//   pentagon   src/sgl/cpu_kernels/pentagon.h:2-17     (pentagon_edge_warp_nested.cuh:3)
//   diamond    src/sgl/cpu_kernels/diamond.h:1-14      (the LISTING form, diamond_nested.cuh:4-31: materialise
//              S = N(v0) ^ N(v1), then for every v2 in S count the v3 in S below it -- count_smaller; the default
//              diamond path counts C(|S|,2) per edge instead, diamond_count.cuh:15-17)
// Task = one symmetry-broken edge (v0,v1), v1 < v0, taken by one wave (waves dequeue chunks of consecutive CSR
// entries); the inner loops are wave-uniform and every set operation is a cooperative 64-lane primitive from
// gm_setops.h (lanes stride the shorter list with coalesced loads, bisect the longer one). Unlike the flattened
// mine_kernel these kernels give a whole wave to each set operation -- they are the parity-first versions.
#include "gm_mine.h"
#include "gm_setops.h"

namespace gm {

template <int PAT>
__global__ __launch_bounds__(256) void sgl_nested_kernel(const SglParams p) {
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63;
  const int wave_slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int *S = p.scratch ? p.scratch + (size_t)wave_slot * (size_t)p.max_deg : nullptr;  // house: N(v0) ^ N(v1)
  unsigned long long cnt = 0;
  for (;;) {
    unsigned q = 0;
    if (lane == 0) q = atomicAdd(p.queue, 1u);
    q = (unsigned)readfirst((int)q);
    if ((long long)q >= p.count) break;
    const long long c = p.first + (long long)q * p.step;
    const long long e0 = c * p.chunk;
    const long long e1 = min((long long)p.g.ne, e0 + p.chunk);
    // row of the first entry: largest v with rp[v] <= e0 (wave-uniform bisection)
    int lo = 0, hi = p.g.nv - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((long long)rp[mid] <= e0) lo = mid; else hi = mid - 1;
    }
    int v0 = lo;
    for (long long e = e0; e < e1; ++e) {
      while ((long long)rp[v0 + 1] <= e) ++v0;
      const int v1 = col[e];
      if (v1 >= v0) {  // symmetry break: rows ascend, nothing left in this row
        e = (long long)rp[v0 + 1] - 1;
        continue;
      }
      const int r0 = rp[v0], a0 = rp[v0 + 1] - r0;
      const int *A0 = col + r0;
      const int idx1 = (int)(e - r0);  // neighbours of v0 below v1 are A0[0 .. idx1)
      const int *B1 = col + rp[v1];
      const int b1 = rp[v1 + 1] - rp[v1];
      if (PAT == SGL_DIAMOND) {
        const int n = wave_intersect_set(A0, a0, B1, b1, S);
        wave_sync();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        for (int s = 0; s < n; ++s) cnt += wave_count_smaller(S[s], S, n);  // v2 = S[s], the v3 < v2 of S
        wave_sync();
      } else if (PAT == SGL_RECTANGLE) {
        for (int j = 0; j < idx1; ++j) {
          const int v2 = A0[j];
          cnt += wave_intersect_num_upper(B1, b1, col + rp[v2], rp[v2 + 1] - rp[v2], v0);
        }
      } else if (PAT == SGL_HOUSE) {
        const int n = wave_intersect_set(A0, a0, B1, b1, S);
        wave_sync();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        for (int s = 0; s < n; ++s) {
          const int v2 = S[s];
          for (int t = 0; t < b1; ++t) {
            const int v3 = B1[t];
            if (v3 == v0 || v3 == v2) continue;
            cnt += wave_intersect_num_except2(A0, a0, col + rp[v3], rp[v3 + 1] - rp[v3], v1, v2);
          }
        }
        wave_sync();
      } else {  // SGL_PENTAGON
        for (int j = 0; j < idx1; ++j) {
          const int v2 = A0[j];
          const int *B2 = col + rp[v2];
          const int b2 = rp[v2 + 1] - rp[v2];
          for (int t = 0; t < b2; ++t) {
            const int v3 = B2[t];
            if (v3 >= v0) break;
            if (v3 == v1) continue;
            cnt += wave_intersect_num_upper_except(B1, b1, col + rp[v3], rp[v3 + 1] - rp[v3], v0, v2);
          }
        }
      }
    }
  }
  const unsigned long long s0 = wave_sum_u64(cnt);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_sgl_nested(int pat, const SglParams &p, int grid_blocks, hipStream_t stream) {
  dim3 grid((unsigned)grid_blocks), block(256);
  switch (pat) {
    case SGL_DIAMOND: hipLaunchKernelGGL(sgl_nested_kernel<SGL_DIAMOND>, grid, block, 0, stream, p); break;
    case SGL_RECTANGLE: hipLaunchKernelGGL(sgl_nested_kernel<SGL_RECTANGLE>, grid, block, 0, stream, p); break;
    case SGL_HOUSE: hipLaunchKernelGGL(sgl_nested_kernel<SGL_HOUSE>, grid, block, 0, stream, p); break;
    case SGL_PENTAGON: hipLaunchKernelGGL(sgl_nested_kernel<SGL_PENTAGON>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_sgl_kernel() {}
void gm_touch_sgl() { hipLaunchKernelGGL(gm_touch_sgl_kernel, dim3(1), dim3(1), 0, 0); }

