// gm_mine_wide.hip -- the big-LDS workgroup classes of the mining kernel for the symmetric-graph patterns (diamond, 3-motif,
// the per-edge sums of 4-motif): MineCfg<PAT, 1> stages a whole row of 3073..8191 entries in 32 KB, MineCfg<PAT, 2> a whole row
// of up to 24576 entries in 96 KB of the CU's 160 KB LDS (gm_chunk.h). Same process_chunk, same flattened passes -- only the
// LDS budget and the waves per workgroup differ, so that these rows are searched in LDS instead of through a dense bitmap in
// HBM. Since the hashed-row kernels (gm_hrow.hip) these sorted-copy classes are the fallback for id spaces too wide for their
// 14-bit remainders (nv > 2^25 / 2^27) and the A/B baseline (tune[6] & 0x400000). (reference kernels being replaced: src/motif/gpu_kernels/motif3_edge_warp.cuh:2-23, src/sgl/gpu_kernels/diamond_count.cuh:3-21)
#include "gm_chunk.h"

namespace gm {

size_t mine_wide_lds_bytes(int cls) { return cls == 2 ? sizeof(BlockLds<PAT_DIAMOND, 2>) : sizeof(BlockLds<PAT_DIAMOND, 1>); }
int mine_wide_threads(Pattern pat, int cls) { return GM_WAVE * (cls == 2 ? MineCfg<PAT_DIAMOND, 2>::waves : mid_waves_of(pat)); }

hipError_t launch_mine_wide(Pattern pat, int cls, const MineParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(BlockLds<PAT_DIAMOND, 2>) <= 163840, "class 2 must fit the 160 KB of one CU");
  static_assert(sizeof(BlockLds<PAT_DIAMOND, 1>) * 2 <= 163840, "two class-1 workgroups per CU");
  const dim3 grid((unsigned)grid_blocks), block((unsigned)mine_wide_threads(pat, cls));
#define GM_WIDE_CASE(P)                                                                          \
  case P:                                                                                        \
    static_assert(MineCfg<P, 1>::waves == mid_waves_of(P), "launch width = the kernel's");         \
    if (cls == 2) hipLaunchKernelGGL((mine_kernel<P, 2>), grid, block, 0, stream, p);             \
    else hipLaunchKernelGGL((mine_kernel<P, 1>), grid, block, 0, stream, p);                      \
    break;
  switch (pat) {
    GM_WIDE_CASE(PAT_DIAMOND)
    GM_WIDE_CASE(PAT_MOTIF3)
    GM_WIDE_CASE(PAT_MOTIF4E)
    default: return hipErrorInvalidValue;
  }
#undef GM_WIDE_CASE
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_mine_wide_kernel() {}
void gm_touch_mine_wide() { hipLaunchKernelGGL(gm_touch_mine_wide_kernel, dim3(1), dim3(1), 0, 0); }

