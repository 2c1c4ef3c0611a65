// gm_tct.hip -- triangle counting with the SHORTER list streamed: |N+(u) ^ N+(v)| is symmetric in u and v, so every DAG edge is
// hosted by the endpoint with the LONGER out-list (its row is the one staged in LDS) and the other endpoint's list is streamed.
// The reference streams N+(v) of every out-edge u -> v (src/triangle/gpu_kernels/bs_warp_edge.cuh:2-18: one warp per edge, the
// shorter list is searched in the longer); the chunked kernel of gm_chunk.h staged N+(u) and streamed N+(v) -- on R-MAT-22
// 12.6 G keys -- whichever was longer, unless the gap was large enough for the bisection-in-HBM pass.  Streaming min(d+(u),
// d+(v)) per edge is 5.0 G keys.  The tasks of a vertex are then its out-edges to shorter-or-equal lists AND its in-edges from
// shorter lists: a second CSR ("task lists", built once per graph on the device: gm_api.hip ensure_tasklists) whose entries are
// the descriptors {rp[partner], d+(partner)} of the lists to stream.  A chunk = consecutive vertices whose DAG rows fit the
// stage of 1024 entries (2048 where the longest row needs it); its tasks = the task-list entries of the same vertices.
#include "gm_flat.h"

namespace gm {

template <int STAGE>
struct alignas(16) TctLds {
  int stage[STAGE];                // the chunk's DAG rows (the stationary side)
  unsigned fbits[kFilterWords];    // hashed membership filter of (local row, id)
  int rpl[kMaxChunkVerts + 1];     // row offsets of the chunk's DAG rows (global entry indices)
  int trpl[kMaxChunkVerts + 1];    // row offsets of its task lists
  WaveLdsLean w[kWavesPerBlock];
  int next_batch;
  unsigned queue_pos;
  int pad_[2];
};

__device__ __forceinline__ int local_row(const int *rpl, const int nvl, const int e) {  // largest i with rpl[i] <= e
  int lo = 0, hi = nvl - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int STAGE>
__global__ __launch_bounds__((kWavesPerBlock * GM_WAVE), (STAGE <= 1024 ? 8 : 6))
void tct_kernel(const MineParams p) {
  __shared__ TctLds<STAGE> B;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x, nthreads = kWavesPerBlock * GM_WAVE;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int *__restrict__ trp = p.g.trp;
  const int2 *__restrict__ tdesc = p.g.tdesc;
  WaveLdsLean &L = B.w[wave];
  unsigned long long c0 = 0;
  for (;;) {
    if (tid == 0) B.queue_pos = atomicAdd(p.queue, (unsigned)p.grab);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const unsigned qe = min(q + (unsigned)p.grab, (unsigned)p.count);
    for (unsigned ci = q; ci < qe; ++ci) {
      const size_t pos = (size_t)p.first + (size_t)ci * (size_t)p.step;
      const size_t cid = p.order ? (size_t)p.order[pos] : pos;
      const ChunkRec r = p.chunks[cid];
      const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
      const int eb = r.e_begin, nel = r.e_end - r.e_begin;
      // ---- workgroup: stage the DAG rows, build the filter ------------------------------------------------------------
      for (int i = tid; i <= nvl; i += nthreads) {
        B.rpl[i] = rp[ub + i];
        B.trpl[i] = trp[ub + i];
      }
      for (int i = tid; i < kFilterWords; i += nthreads) B.fbits[i] = 0u;
      for (int i = tid; i < nel; i += nthreads) B.stage[i] = col[eb + i];
      if (tid == 0) B.next_batch = 0;
      __syncthreads();
      for (int i = tid; i < nel; i += nthreads) {
        const int lo = local_row(B.rpl, nvl, eb + i);
        const unsigned h = filter_hash<kFilterLog2>(B.stage[i], filter_salt(lo));
        atomicOr(&B.fbits[h >> 5], 1u << (h & 31u));
      }
      __syncthreads();
      // ---- waves: batches of 64 tasks ---------------------------------------------------------------------------------
      const int tb = B.trpl[0], ntask = B.trpl[nvl] - tb;
      // (smaller batches for parts with few tasks -- what gm_cbuild.hip does for the thin task lists of a rank's share -- were measured
      // here and lose: R-MAT-22 5.29 -> 5.51 ms on one GPU, 0.97 -> 0.97 ms per rank of eight; profiles/r03/ab_share_scaling.txt)
      constexpr int bsz = GM_WAVE;
      for (;;) {
        int bi = 0;
        if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
        bi = readfirst(bi) * r.nparts + r.part;
        const int t0 = bi * bsz;
        if (t0 >= ntask) break;
        const bool valid = lane < bsz && t0 + lane < ntask;
        const int te = tb + min(t0 + lane, ntask - 1);
        const int2 d = tdesc[te];                 // {rp[partner], d+(partner)}: the list to stream, coalesced
        const int lo = local_row(B.trpl, nvl, te);  // the host row of this task
        const int ru = B.rpl[lo], a = B.rpl[lo + 1] - ru;
        const bool act = valid && d.y > 0 && a > 0;
        unsigned cnt = 0;
        auto found = [&](bool f, int, int, int, int, int) { cnt += f ? 1u : 0u; };
        flat_pass_filtered<kFilterLog2>(L, B.stage, B.fbits, col, lane, act ? d.y : 0, d.x, (ru - eb) | (int)(filter_salt(lo) << 16), a,
                                        p.flags, found);
        c0 += (unsigned long long)cnt;
      }
      __syncthreads();  // the stage is rewritten by the next chunk
    }
  }
  const unsigned long long s0 = wave_sum_u64(c0);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

// stage: 1024 (20 KB of LDS, eight workgroups per CU) or 2048 entries (24 KB, six) -- the longest DAG row must fit
int tct_per_cu(int stage) { return stage <= 1024 ? 8 : 6; }
hipError_t launch_tct(const MineParams &p, int stage, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(TctLds<1024>) * 8 <= 163840, "eight workgroups per CU");
  static_assert(sizeof(TctLds<kTctStageMax>) * 6 <= 163840, "six workgroups per CU");
  if (p.g.trp == nullptr || p.g.tdesc == nullptr) return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_blocks), block(kWavesPerBlock * GM_WAVE);
  if (stage <= 1024) hipLaunchKernelGGL((tct_kernel<1024>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((tct_kernel<kTctStageMax>), grid, block, 0, stream, p);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_tct_kernel() {}
void gm_touch_tct() { hipLaunchKernelGGL(gm_touch_tct_kernel, dim3(1), dim3(1), 0, 0); }

