// gm_cbuild.hip -- k-clique (k = 4), first DFS level RE-HOSTED: every DAG edge u -> v is a task of the endpoint with the longer
// out-list, the other list is streamed against it, and the finished row of u's adjacency bit-matrix is stored into the matrix arena
// (see "k-clique, level 1 re-hosted" in gm_mine.h).  The reference's kernel re-intersects N+(v0) ^ N+(v1) per edge with the shorter
// list searched in the longer (src/clique/gpu_kernels/clique4_warp_edge.cuh:19-27) and never keeps the result; round 2 of this
// library streamed N+(v) of every edge whichever was longer (12.6 vs 5.0 G keys on the LiveJournal stand-in, gm_tct.hip).
//
// Kernel = the shorter-list-streams triangle kernel (gm_tct.hip) with positions: a chunk is a run of consecutive vertices whose DAG
// rows fit the LDS stage, staged behind the salted bit filter; its tasks are the task-list entries of those vertices, 64 per batch;
// the filtered pass carries the STREAM INDEX of every candidate through its queue (flat_pass_filtered<..., IDX>), so a match is
// either bit `position in the staged row` (type A) or bit `bit_off + stream index` (type B) of the task's row.  Rows are built in a
// per-wave LDS buffer of kCbRowBuf words -- a batch is processed in sub-batches of as many tasks as fit -- and stored to the arena by
// the wave that built them: no workgroup barrier, no device atomics.
#include "gm_flat.h"

namespace gm {

template <int STAGE>
struct alignas(16) CBuildLds {
  int stage[STAGE];                // the chunk's DAG rows (the stationary side; first member: the bisection may read past a row)
  unsigned fbits[kFilterWords];    // hashed membership filter of (local row, id)
  int rpl[kMaxChunkVerts + 1];     // row offsets of the chunk's DAG rows (global entry indices)
  int trpl[kMaxChunkVerts + 1];    // row offsets of its task lists
  unsigned rows[kWavesPerBlock][kCbRowBuf];
  WaveLdsIdx w[kWavesPerBlock];
  int next_batch;
  unsigned queue_pos;
  int pad_[2];
};

__device__ __forceinline__ int cb_local_row(const int *rpl, const int nvl, const int e) {  // largest i with rpl[i] <= e
  int lo = 0, hi = nvl - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int STAGE>
__global__ __launch_bounds__((kWavesPerBlock * GM_WAVE), (STAGE <= 1024 ? 6 : 5))
void cbuild_kernel(const CBuildParams p) {
  __shared__ CBuildLds<STAGE> B;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x, nthreads = kWavesPerBlock * GM_WAVE;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int *__restrict__ trp = p.trp;
  const int4 *__restrict__ tasks = reinterpret_cast<const int4 *>(p.tasks);
  unsigned *__restrict__ mat = p.mat;
  WaveLdsIdx &L = B.w[wave];
  unsigned *rb = B.rows[wave];
  for (;;) {
    if (tid == 0) B.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const ChunkRec r = p.chunks[p.order ? p.order[q] : (int)q];
    const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
    const int eb = r.e_begin, nel = r.e_end - r.e_begin;
    const int tb = trp[ub], ntask = trp[ub + nvl] - tb;
    if (ntask == 0) { __syncthreads(); continue; }  // (these vertices host nothing for this rank: nothing to stage)
    // ---- workgroup: stage the DAG rows, build the filter ----------------------------------------------------------------
    for (int i = tid; i <= nvl; i += nthreads) {
      B.rpl[i] = rp[ub + i];
      B.trpl[i] = trp[ub + i];
    }
    for (int i = tid; i < kFilterWords; i += nthreads) B.fbits[i] = 0u;
    for (int i = tid; i < nel; i += nthreads) B.stage[i] = col[eb + i];
    if (tid == 0) B.next_batch = 0;
    __syncthreads();
    for (int i = tid; i < nel; i += nthreads) {
      const int lo = cb_local_row(B.rpl, nvl, eb + i);
      const unsigned h = filter_hash<kFilterLog2>(B.stage[i], filter_salt(lo));
      atomicOr(&B.fbits[h >> 5], 1u << (h & 31u));
    }
    __syncthreads();
    // ---- waves: batches of 64 tasks, sub-batches of as many rows as the wave's row buffer holds -----------------------------
    // (a chunk with few tasks -- the share of one rank of eight holds ~125 per chunk -- takes smaller batches, so that all four waves
    // get some: with 64 two waves of every workgroup idled, the build of a 1/8 share ran at 0.68 of its ideal)
    const int per_part = (ntask + r.nparts - 1) / r.nparts;
    const int bsz = per_part >= 8 * GM_WAVE ? GM_WAVE : (per_part >= 4 * GM_WAVE ? 32 : 16);
    for (;;) {
      int bi = 0;
      if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
      bi = readfirst(bi) * r.nparts + r.part;
      const int t0 = bi * bsz;
      if (t0 >= ntask) break;
      const int nvalid = min(bsz, ntask - t0);
      const bool valid = lane < nvalid;
      const int te = tb + min(t0 + lane, ntask - 1);
      const int4 T = tasks[te];                       // coalesced, 16 B per lane
      const int lo = cb_local_row(B.trpl, nvl, te);   // the host row of this task
      const int ru = B.rpl[lo], a = B.rpl[lo + 1] - ru;
      const unsigned fl = (unsigned)T.w;
      const int words = valid ? (int)((fl >> 20) & 127u) : 0;
      const int bit_off = (int)((fl >> 8) & 4095u);
      const bool type_b = (fl >> 31) != 0u;
      const unsigned long long off = ((unsigned long long)(fl & 255u) << 32) | (unsigned long long)(unsigned)T.z;
      const int incl = wave_incl_scan_add(words);
      int start = 0, consumed = 0;  // wave-uniform: first lane / first word of the current sub-batch
      while (start < nvalid) {
        const unsigned long long fit = __ballot(lane >= start && valid && incl - consumed <= kCbRowBuf);
        const int cnt = __popcll(fit);  // >= 1: a row is at most 64 words
        const bool in_sub = lane >= start && lane < start + cnt;
        const int row0 = incl - words - consumed;  // this lane's row inside the buffer
        const int used = readlane(incl, start + cnt - 1) - consumed;
        for (int i = lane; i < used; i += GM_WAVE) rb[i] = 0u;
        L.meta[lane] = (row0 & 4095) | (bit_off << 12) | (type_b ? (int)0x80000000u : 0);
        wave_sync();
        auto found = [&](bool f, int owner, int sidx, int pos, int, int) {
          if (!f) return;
          const int m = L.meta[owner];
          const int bit = (m < 0) ? ((m >> 12) & 4095) + sidx : pos;  // type B: stream index; type A: position in the staged row
          atomicOr(&rb[(m & 4095) + (bit >> 5)], 1u << (bit & 31));
        };
        flat_pass_filtered<kFilterLog2, false, true>(L, B.stage, B.fbits, col, lane, (in_sub && a > 0) ? T.y : 0, T.x,
                                                      (ru - eb) | (int)(filter_salt(lo) << 16), a, p.flags, found);
        wave_sync();
        // store the finished rows (all of them: a row without a match is a row of zeros)
        const int maxw = wave_max_nonneg(in_sub ? words : 0);
        if (maxw <= 8) {  // short rows: every lane stores its own (consecutive tasks are mostly consecutive rows of one matrix)
          for (int i = 0; i < maxw; ++i)
            if (in_sub && i < words) mat[off + (unsigned long long)i] = rb[row0 + i];
        } else {
          for (int rr = start; rr < start + cnt; ++rr) {  // wave-uniform: one coalesced store per row
            const int w_r = readlane(words, rr), r0 = readlane(row0, rr);
            const unsigned long long o_r = ((unsigned long long)(unsigned)readlane((int)(off >> 32), rr) << 32) | (unsigned)readlane((int)(unsigned)off, rr);
            if (lane < w_r) mat[o_r + (unsigned long long)lane] = rb[r0 + lane];
          }
        }
        wave_sync();
        consumed += used;
        start += cnt;
      }
    }
    __syncthreads();  // the stage is rewritten by the next chunk
  }
}

int cbuild_per_cu(int stage) { return stage <= 1024 ? 6 : 5; }
hipError_t launch_cbuild(const CBuildParams &p, int stage, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(CBuildLds<1024>) * 6 <= 163840, "six workgroups per CU");
  static_assert(sizeof(CBuildLds<kCbMaxDeg>) * 5 <= 163840, "five workgroups per CU");
  static_assert(kCbMaxDeg <= 4096 && kCbRowBuf <= 4096, "bit offsets and row offsets are 12-bit fields");
  if (p.trp == nullptr || p.tasks == nullptr || p.mat == nullptr) return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_blocks), block(kWavesPerBlock * GM_WAVE);
  if (stage <= 1024) hipLaunchKernelGGL((cbuild_kernel<1024>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((cbuild_kernel<kCbMaxDeg>), grid, block, 0, stream, p);
  return hipGetLastError();
}

// ---- second level of the NARROW vertices ---------------------------------------------------------------------------------------
// The matrices of a narrow chunk's vertices are contiguous in the arena (vertex order): one coalesced copy into LDS, then one thread
// per row i:  sum_{j in M_i} popc(M_i & M_j)  (clique4_warp_edge.cuh:22-27 on bit rows) -- with a topological numbering M_j has no bit
// at or below j, so the words below j / 32 are skipped.
struct alignas(16) SmallLds {
  unsigned bits[kBitWords];
  int rpl[kMaxChunkVerts + 1];
  int boff[kMaxChunkVerts + 1];  // LDS word offset of every vertex' matrix
  unsigned queue_pos;
  int pad_[3];
};

__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE, 8) void clique_small_kernel(const CliqueSmallParams p) {
  __shared__ SmallLds S;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1);
  constexpr int NT = kWavesPerBlock * GM_WAVE;
  unsigned long long tot = 0;
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = S.queue_pos;
    if (q >= (unsigned)p.count) break;
    const size_t pos = (size_t)p.first + (size_t)q * (size_t)p.step;
    const ChunkRec r = p.chunks[p.order ? (size_t)p.order[pos] : pos];
    const int ub = r.u_begin, nvl = r.u_end - r.u_begin, eb = r.e_begin, nel = r.e_end - r.e_begin;
    const unsigned long long b0 = p.base[ub];
    const int nwords = (int)(p.base[ub + nvl] - b0);  // <= kBitWords: the chunk table was cut for that
    if (nwords == 0) { __syncthreads(); continue; }
    for (int i = tid; i <= nvl; i += NT) {
      S.rpl[i] = p.rp[ub + i];
      S.boff[i] = (int)(p.base[ub + i] - b0);
    }
    for (int i = tid; i < nwords; i += NT) S.bits[i] = p.mat[b0 + (unsigned long long)i];
    __syncthreads();
    unsigned c = 0;
    for (int le = tid; le < nel; le += NT) {
      const int lo = cb_local_row(S.rpl, nvl, eb + le);
      const int d = S.rpl[lo + 1] - S.rpl[lo];
      if (S.boff[lo + 1] == S.boff[lo]) continue;  // d < kCbMinDeg: no matrix
      const int s = (d + 31) >> 5, i = eb + le - S.rpl[lo];
      const unsigned *M = S.bits + S.boff[lo];
      const unsigned *Mi = M + i * s;
      for (int w = 0; w < s; ++w) {
        unsigned x = Mi[w];
        while (x) {
          const int bit = __ffs((int)x) - 1;
          x &= x - 1;
          const unsigned *Mj = M + (w * 32 + bit) * s;
          for (int w2 = p.topo ? w : 0; w2 < s; ++w2) c += (unsigned)__popc(Mi[w2] & Mj[w2]);  // (topological: M_j has no bit below word w)
        }
      }
    }
    tot += (unsigned long long)c;
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64(tot);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_clique_small(const CliqueSmallParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(clique_small_kernel, dim3((unsigned)grid_blocks), dim3(kWavesPerBlock * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_cbuild_kernel() {}
void gm_touch_cbuild() { hipLaunchKernelGGL(gm_touch_cbuild_kernel, dim3(1), dim3(1), 0, 0); }

