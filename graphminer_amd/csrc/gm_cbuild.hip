// gm_cbuild.hip -- k-clique (k = 4), first DFS level RE-HOSTED: every DAG edge u -> v is a task of the endpoint with the longer
// out-list, the other list is streamed against it, and the finished row of u's adjacency bit-matrix is stored into the matrix arena
// (see "k-clique, level 1 re-hosted" in gm_mine.h).  The reference's kernel re-intersects N+(v0) ^ N+(v1) per edge with the shorter
// list searched in the longer (src/clique/gpu_kernels/clique4_warp_edge.cuh:19-27) and never keeps the result; round 2 of this
// library streamed N+(v) of every edge whichever was longer (12.6 vs 5.0 G keys on the LiveJournal stand-in, gm_tch.hip).
//
// Kernel = the shorter-list-streams triangle kernel with positions: a chunk is a run of consecutive vertices whose DAG rows fit the LDS
// stage; its tasks are the task-list entries of those vertices, 64 per batch.  A match is either bit `position in the host's row`
// (type A) or bit `bit_off + index in the streamed list` (type B) of the task's row.  Rows are built in a per-wave LDS buffer of
// kCbRowBuf words -- a batch is processed in sub-batches of as many tasks as fit -- and stored to the arena by the wave that built
// them: no workgroup barrier, no device atomics.
//
// The chunk's rows are ONE HASHED SET of (row, id) -> position in LDS (gm_hset.h; round 3 -- before: sorted copy + bit filter +
// candidate queue + bisection, 80 VALU per 64 streamed keys with 30 % of the keys of the com-Orkut stand-in going through all of it).
#include "gm_hset.h"

namespace gm {

#ifndef GM_CB_TILES
#define GM_CB_TILES 4
#endif
constexpr int kCbTiles = GM_CB_TILES;  // 64-key tiles in flight per wave (1024-entry stage)
// The 2048-entry stage (61 KB of set + wave scratch: two workgroups per CU) is bound by the latency of each wave's own chain load -> hash -> bucket ->
// compare, like the blocked gather (profiles/r06/ab_clique4_blocked_gather.txt): TWELVE waves per workgroup with two tiles in flight (80 registers,
// six waves per SIMD) instead of eight with four (128 registers, four per SIMD): 22.69 -> 22.16 ms for the whole pattern; sixteen waves at 64
// registers spill (22.96).  profiles/r06/ab_clique4_cbuild_occupancy.txt
#ifndef GM_CB_TILES_BIG
#define GM_CB_TILES_BIG 2
#endif

template <int STAGE, int WAVES>
struct alignas(16) CBuildLds {
  HsTable<STAGE> set;
  int trpl[kMaxChunkVerts + 1];    // row offsets of the chunk's task lists
  unsigned rows[WAVES][kCbRowBuf];
  HsWave<STAGE> w[WAVES];                 // (while the set is built: the fill counters of its buckets)
  // the chunk being worked on and the NEXT one (round 6): queue position, record, start and number of its tasks -- fetched by thread 0 while the
  // other waves stream the current chunk's batches; the chain dequeue -> order -> record -> task bounds was four dependent round trips per chunk
  ChunkRec grec[2];
  unsigned gq[2];
  int gtb[2], gnt[2];
  int next_batch;
  int pad_[3];
};

template <int STAGE, int WAVES>
__global__ __launch_bounds__(WAVES * GM_WAVE) __attribute__((amdgpu_waves_per_eu(STAGE <= 1024 ? 1 : (WAVES * 2) / 4)))
void cbuild_kernel(const CBuildParams p) {
  __shared__ CBuildLds<STAGE, WAVES> B;
  using H = HsHash<STAGE>;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int tid = threadIdx.x;
  constexpr int nthreads = WAVES * GM_WAVE;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int *__restrict__ trp = p.trp;
  const int4 *__restrict__ tasks = reinterpret_cast<const int4 *>(p.tasks);
  unsigned *__restrict__ mat = p.mat;
  HsWave<STAGE> &L = B.w[wave];
  unsigned *rb = B.rows[wave];
  auto fetch = [&](const int slot) {  // thread 0: dequeue + the chunk's record and task bounds into slot `slot`
    const unsigned qq = atomicAdd(p.queue, 1u);
    B.gq[slot] = qq;
    if (qq < (unsigned)p.count) {
      const ChunkRec rr = p.chunks[p.order ? p.order[qq] : (int)qq];
      const int t0 = trp[rr.u_begin];
      B.grec[slot] = rr;
      B.gtb[slot] = t0;
      B.gnt[slot] = trp[rr.u_end] - t0;
    }
  };
  if (tid == 0) fetch(0);
  __syncthreads();
  for (int it = 0;; ++it) {
    const int cur = it & 1;
    const unsigned q = B.gq[cur];
    if (q >= (unsigned)p.count) break;
    const ChunkRec r = B.grec[cur];
    const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
    const int eb = r.e_begin, nel = r.e_end - r.e_begin;
    const int tb = B.gtb[cur], ntask = B.gnt[cur];
    if (ntask == 0) {  // (workgroup-uniform: these vertices host nothing for this rank -- nothing to stage)
      if (tid == 0) fetch(cur ^ 1);
      __syncthreads();
      continue;
    }
    // ---- workgroup: the chunk's DAG rows into the set -------------------------------------------------------------------
    for (int i = tid; i <= nvl; i += nthreads) B.trpl[i] = trp[ub + i];
    if (tid == 0) B.next_batch = 0;
    const bool fallback = hs_build<STAGE, nthreads>(B.set, reinterpret_cast<unsigned *>(&B.w[0]), rp, col, ub, nvl, eb, nel,
                                                     (p.flags & (1 << 22)) != 0, tid, B.trpl);  // (ends with a barrier; only the rows that host a task)
    if (tid == 0) fetch(cur ^ 1);  // (its wave joins the batches when the three loads are back; the others have started)
    // ---- waves: batches of 64 tasks, sub-batches of as many rows as the wave's row buffer holds -----------------------------
    // (a chunk with few tasks -- the share of one rank of eight holds ~125 per chunk -- takes smaller batches, so that all waves
    // get some: with 64 two waves of every workgroup idled, the build of a 1/8 share ran at 0.68 of its ideal)
    const int per_part = (ntask + r.nparts - 1) / r.nparts;
    const int bsz = per_part >= 2 * nthreads ? GM_WAVE : (per_part >= nthreads ? 32 : 16);
    for (;;) {
      int bi = 0;
      if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
      bi = readfirst(bi) * r.nparts + r.part;
      const int t0 = bi * bsz;
      if (t0 >= ntask) break;
      const int nvalid = min(bsz, ntask - t0);
      const bool valid = lane < nvalid;
      const int te = tb + min(t0 + lane, ntask - 1);
      const int4 Tk = tasks[te];                      // coalesced, 16 B per lane
      const int lo = hs_local_row(B.trpl, nvl, te);   // the host row of this task
      const int ru = B.set.rpl[lo], a = B.set.rpl[lo + 1] - ru;
      const unsigned fl = (unsigned)Tk.w;
      const int words = valid ? (int)((fl >> 20) & 127u) : 0;
      const int bit_off = (int)((fl >> 8) & 4095u);
      const bool type_b = (fl >> 31) != 0u;
      const unsigned long long off = ((unsigned long long)(fl & 255u) << 32) | (unsigned long long)(unsigned)Tk.z;
      const int incl_w = wave_incl_scan_add(words);
      int start = 0, consumed = 0;  // wave-uniform: first lane / first word of the current sub-batch
      while (start < nvalid) {
        const unsigned long long fit = __ballot(lane >= start && valid && incl_w - consumed <= kCbRowBuf);
        const int cnt = __popcll(fit);  // >= 1: a row is at most 64 words
        const bool in_sub = lane >= start && lane < start + cnt;
        const int row0 = incl_w - words - consumed;  // this lane's row inside the buffer
        const int used = readlane(incl_w, start + cnt - 1) - consumed;
        for (int i = lane; i < used; i += GM_WAVE) rb[i] = 0u;
        // A match sets bit (row0 * 32 + b) of the wave's row buffer, b = the match's position in the host row (type A) or bit_off + its
        // index in the streamed list (type B: = bit_off - list start + its index in col).  The task's two words fold that into one
        // v_bfi + one add per match: word = row0 * 32 (+ bit_off - list start for type B), word2 = all ones for type B, else 0.
        const int base_l = row0 * 32 + (type_b ? bit_off - Tk.x : 0);
        const int sel_l = type_b ? -1 : 0;
        wave_sync();
        auto hit = [&](const unsigned long long hm, const int base, const int sel, const unsigned at, const int kidx, const bool) {
          if (hm == 0ull) return;  // wave-uniform
#ifndef GM_CB_ABLATE_HITS  // (A/B builds: what the row-buffer writes cost; counts wrong)
          if (__builtin_amdgcn_inverse_ballot_w64(hm)) {
            const unsigned bit = (unsigned)base + (((unsigned)kidx & (unsigned)sel) | (at & ~(unsigned)sel));
            atomicOr(&rb[bit >> 5], 1u << (bit & 31u));
          }
#endif
        };
        auto hit1 = [&](const int base, const int sel, const int at, const int kidx) {
          if (lane == 0) {
            const unsigned bit = (unsigned)base + (((unsigned)kidx & (unsigned)sel) | ((unsigned)at & ~(unsigned)sel));
            atomicOr(&rb[bit >> 5], 1u << (bit & 31u));
          }
        };
        hs_pass<STAGE, (STAGE <= 1024 ? kCbTiles : GM_CB_TILES_BIG)>(B.set, L, col, fallback, lane, (in_sub && a > 0) ? Tk.y : 0, Tk.x, H::salt(lo), ru - eb, a, base_l, sel_l, hit, hit1);
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
    __syncthreads();  // the set is rewritten by the next chunk
  }
}

// workgroups of 4 waves on the 1024-entry stage (32 KB of LDS), of GM_CB_WAVES_BIG on the 2048-entry one
// (R-MAT-22 ef 28, 4 / 6 / 8 waves: 68.5 / 70.6 / 66.7 ms for the whole pattern)
#ifndef GM_CB_WAVES_BIG
#define GM_CB_WAVES_BIG 12
#endif
constexpr int kCbWavesBig = GM_CB_WAVES_BIG;
int cbuild_per_cu(int stage) {
  return (int)(163840 / (stage <= 1024 ? sizeof(CBuildLds<1024, 4>) : sizeof(CBuildLds<kCbMaxDeg, kCbWavesBig>)));
}
hipError_t launch_cbuild(const CBuildParams &p, int stage, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(CBuildLds<1024, 4>) * 3 <= 163840, "three workgroups per CU at least (the registers allow no more)");
  static_assert(sizeof(CBuildLds<kCbMaxDeg, kCbWavesBig>) * 2 <= 163840, "two workgroups per CU");
  static_assert(sizeof(HsWave<kCbMaxDeg>) * 4 >= (size_t)kCbMaxDeg * 2, "fill counters alias the wave scratch");
  static_assert(kCbMaxDeg <= 2048 && kCbRowBuf <= 4096, "positions are 11-bit fields of a slot, bit offsets and row offsets 12-bit fields");
  if (p.trp == nullptr || p.tasks == nullptr || p.mat == nullptr) return hipErrorInvalidValue;
  const dim3 grid((unsigned)grid_blocks);
  if (stage <= 1024) hipLaunchKernelGGL((cbuild_kernel<1024, 4>), grid, dim3(4 * GM_WAVE), 0, stream, p);
  else hipLaunchKernelGGL((cbuild_kernel<kCbMaxDeg, kCbWavesBig>), grid, dim3(kCbWavesBig * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

// ---- second level of the NARROW vertices ---------------------------------------------------------------------------------------
// The matrices of a narrow chunk's vertices are contiguous in the arena (vertex order): one coalesced copy into LDS, then one thread
// per row i:  sum_{j in M_i} popc(M_i & M_j)  (clique4_warp_edge.cuh:22-27 on bit rows) -- with a topological numbering M_j has no bit
// at or below j, so the words below j / 32 are skipped.
// Round 6: the kernel was one dependent chain per chunk -- dequeue atomic -> chunk record -> matrix offsets -> matrix words -> barrier ->
// count (waves waiting 0.78 of their cycles, every unit below 0.4: profiles/r05/clique4_rmat22ef28_pmc_summary.txt).  Now a workgroup takes
// kSmallGrab chunks per atomic, eight threads fetch their records and offsets side by side, and the words of chunk k + 1 are requested (into
// registers, then the OTHER of two LDS buffers) before chunk k is counted.
constexpr int kSmallGrab = 8;
struct SmallInfo {
  int ub, nvl, eb, nel, nwords, pad_;
  unsigned long long b0;
};
struct alignas(16) SmallLds {
  unsigned bits[2][kBitWords];
  int rpl[2][kMaxChunkVerts + 1];
  int boff[2][kMaxChunkVerts + 1];  // LDS word offset of every vertex' matrix
  SmallInfo info[kSmallGrab];
  unsigned queue_pos;
  int pad_[3];
};

__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE, 8) void clique_small_kernel(const CliqueSmallParams p) {
  __shared__ SmallLds S;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1);
  constexpr int NT = kWavesPerBlock * GM_WAVE;
  constexpr int kWordsPerThread = (kBitWords + NT - 1) / NT, kVertsPerThread = (kMaxChunkVerts + 1 + NT - 1) / NT;
  unsigned long long tot = 0;
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, (unsigned)kSmallGrab);
    __syncthreads();
    const unsigned q0 = S.queue_pos;
    if (q0 >= (unsigned)p.count) break;
    const int nq = (int)min((unsigned)kSmallGrab, (unsigned)p.count - q0);
    if (tid < nq) {  // the records and offsets of the batch's chunks, side by side
      const size_t pos = (size_t)p.first + (size_t)(q0 + (unsigned)tid) * (size_t)p.step;
      const ChunkRec r = p.chunks[p.order ? (size_t)p.order[pos] : pos];
      SmallInfo x;
      x.ub = r.u_begin;
      x.nvl = r.u_end - r.u_begin;
      x.eb = r.e_begin;
      x.nel = r.e_end - r.e_begin;
      x.b0 = p.base[x.ub];
      x.nwords = (int)(p.base[x.ub + x.nvl] - x.b0);  // <= kBitWords: the chunk table was cut for that
      x.pad_ = 0;
      S.info[tid] = x;
    }
    __syncthreads();
    // registers of the chunk being fetched
    unsigned wv[kWordsPerThread];
    int rv[kVertsPerThread];
    unsigned long long bv[kVertsPerThread];
    auto fetch = [&](const int k) {  // (no waiting here: the values are used by stash)
      const SmallInfo x = S.info[k];
#pragma unroll
      for (int j = 0; j < kVertsPerThread; ++j) {
        const int i = min(tid + j * NT, x.nvl);
        rv[j] = p.rp[x.ub + i];
        bv[j] = p.base[x.ub + i];
      }
      if (x.nwords > 0) {  // (workgroup-uniform; a chunk without a matrix has nothing to read -- and may sit at the arena's very end)
#pragma unroll
        for (int j = 0; j < kWordsPerThread; ++j) wv[j] = p.mat[x.b0 + (unsigned long long)min(tid + j * NT, x.nwords - 1)];
      }
    };
    auto stash = [&](const int k) {
      const SmallInfo x = S.info[k];
      const int b = k & 1;
#pragma unroll
      for (int j = 0; j < kVertsPerThread; ++j) {
        const int i = tid + j * NT;
        if (i <= x.nvl) {
          S.rpl[b][i] = rv[j];
          S.boff[b][i] = (int)(bv[j] - x.b0);
        }
      }
#pragma unroll
      for (int j = 0; j < kWordsPerThread; ++j)
        if (tid + j * NT < x.nwords) S.bits[b][tid + j * NT] = wv[j];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    unsigned c = 0;
    for (int k = 0; k < nq; ++k) {
      if (k + 1 < nq) fetch(k + 1);  // (wave-uniform) in flight while chunk k is counted
      const SmallInfo x = S.info[k];
      const int b = k & 1;
      if (x.nwords > 0) {
        for (int le = tid; le < x.nel; le += NT) {
          const int lo = hs_local_row(S.rpl[b], x.nvl, x.eb + le);
          const int d = S.rpl[b][lo + 1] - S.rpl[b][lo];
          if (S.boff[b][lo + 1] == S.boff[b][lo]) continue;  // d < kCbMinDeg: no matrix
          const int s = (d + 31) >> 5, i = x.eb + le - S.rpl[b][lo];
          const unsigned *M = S.bits[b] + S.boff[b][lo];
          const unsigned *Mi = M + i * s;
          for (int w = 0; w < s; ++w) {
            unsigned xw = Mi[w];
            while (xw) {
              const int bit = __ffs((int)xw) - 1;
              xw &= xw - 1;
              const unsigned *Mj = M + (w * 32 + bit) * s;
              for (int w2 = p.topo ? w : 0; w2 < s; ++w2) c += (unsigned)__popc(Mi[w2] & Mj[w2]);  // (topological: M_j has no bit below word w)
            }
          }
        }
      }
      if (k + 1 < nq) stash(k + 1);  // into the other buffer: nobody reads it (its last readers passed the barrier below an iteration ago)
      __syncthreads();
    }
    tot += (unsigned long long)c;
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

