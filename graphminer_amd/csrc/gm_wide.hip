// gm_wide.hip -- k-clique, the second DFS level of the WIDE vertices, counted from a big-LDS copy of the vertex's adjacency
// bit-matrix (see "k-clique, wide vertices" in gm_mine.h; the matrices are built by cbuild_kernel, gm_cbuild.hip).
//
// MI355X gives a CU 160 KB of LDS. The mining kernel spends 21-32 KB per workgroup so that 5-7 workgroups share a CU -- right
// for the millions of short rows and for BUILDING the matrices (latency-bound streaming: occupancy is what counts), wrong for
// COUNTING the few tens of thousands of wide ones: sum_i sum_{j in M_i} popc(M_i & M_j) touches every row M_j once per set
// bit of every row that points at it (R-MAT-22 ef 28: 6.9 G pairs, ~100 B each), which round 1 served from a per-workgroup
// arena slot in L2 / Infinity Cache through 8 KB LDS tiles (382 GB of reads per launch). Here one workgroup copies the
// whole matrix of a vertex (up to 112 KB) into LDS once -- 4.5 GB of arena reads per launch in total -- and counts from there:
//   * a wave owns a row i: its words sit in registers (broadcast LDS reads), its set-bit positions are expanded into a
//     per-wave list;
//   * LANE-PER-j: each lane takes one j of the list and reads ITS row M_j with 16-byte LDS loads (rows are padded to a
//     stride whose quarter is odd: 16 lanes reading 16 different rows at the same word offset hit 16 different bank groups),
//     ANDs it with the registers, popcounts -- 64 pairs per step, no idle lanes whatever the row width;
//   * rows wider than the LDS budget (d+ > 896) are counted in COLUMN BLOCKS: the block's columns of all rows are copied, the
//     popcounts of every pair are taken over those columns, block after block.
// (reference loop: src/clique/gpu_kernels/clique4_warp_edge.cuh:19-27, which re-intersects the lists from global memory)
#include "gm_flat.h"

namespace gm {

// ---- phase 2 ----------------------------------------------------------------------------------------------------------------

constexpr int kCountMaxQ = 9;        // 16-byte units of a row (block) held in registers: 36 words

template <int WAVES, int WORDS, int LISTCAP>
struct alignas(16) CountLds {
  unsigned bits[WORDS];                    // the (block of the) matrix, rows padded to `ps` words
  unsigned short plist[WAVES][LISTCAP];    // per wave: the set-bit positions of its current row (LISTCAP >= d+ of the class)
  int next_row;
  unsigned queue_pos;
};

// The rows of one (column block of a) matrix resident in LDS: sum_i sum_{j in M_i} popc(M_i & M_j) over the block's columns.
// NQ = 16-byte units per padded row, compile-time: the NQ row reads of a lane are issued back to back and waited for once
// (with a run-time bound every read sat behind its own branch and its own s_waitcnt: 64 pairs took ~9 LDS round trips).
// popc(M_i & M_j) over the 16-byte units KS .. NQ-1 of the two rows (KS: compile time, so that the reads of a lane are still issued
// back to back and waited for once)
template <int NQ, int KS>
__device__ __forceinline__ unsigned pair_popc(const uint4 (&mr)[NQ], const uint4 *__restrict__ rj) {
  unsigned a = 0;
  constexpr int G = 5;  // units requested together: 9 = 5 + 4 keeps the kernel at 4 waves per SIMD without spills
#pragma unroll
  for (int k0 = KS; k0 < NQ; k0 += G) {
    uint4 m[G];
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (k0 + k < NQ) m[k] = rj[k0 + k];
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (k0 + k < NQ)
        a += (unsigned)__popc(mr[k0 + k].x & m[k].x) + (unsigned)__popc(mr[k0 + k].y & m[k].y) +
             (unsigned)__popc(mr[k0 + k].z & m[k].z) + (unsigned)__popc(mr[k0 + k].w & m[k].w);
  }
  return a;
}

// c0: first word of the column block inside the matrix rows, cwb: its words.  topo: the matrix is strictly upper triangular (row j has
// no bit at or below column j), so
//   * a pair (i, j), i < j, has common bits only beyond column j: rows and columns at or beyond the block's last column contribute
//     nothing to this block -- the row loop and the column scan end there;
//   * for a tile of ascending j's the units below the first j's own are zero in every M_j and are skipped (rows of >= 6 units).
// The j's of row i (its set bits) are COMPACTED 64 COLUMNS PER STEP -- lane = column, ballot, v_mbcnt rank, one 16-bit LDS write -- into
// the wave's list, and a tile of 64 pairs runs as soon as 64 are queued: ~8 VALU per 64 columns whatever the bits look like.  (Until
// round 3 a lane expanded the bits of ITS 32-column word in a loop: the loop ran as long as the densest word of the row -- the hub
// columns at the end of a topologically numbered matrix: 16 / 26 / 29 iterations per row in classes S / L / X of R-MAT-22 ef 28 for
// 35 / 164 / 331 pairs, more instructions than the pair counts themselves.)
template <int NQ, bool WHOLE, int LISTCAP>
__device__ __forceinline__ unsigned long long count_block(const unsigned *__restrict__ bits, unsigned short *__restrict__ plist, int *next_row,
                                                          const unsigned *__restrict__ gm, const int d, const int stride, const int lane,
                                                          const int c0, const int cwb, const bool topo) {
  constexpr int ps = 4 * NQ;
  unsigned long long tot = 0;
  unsigned c = 0;
  const int jend = topo ? min(d, (c0 + cwb) * 32) : d;  // rows / columns that can contribute to this block
  // The set bits of row i over ALL columns drive the enumeration of j. WHOLE (the block is the whole matrix): they are read
  // from LDS. Column blocks: from the arena (L2), and the read of the NEXT row is issued before the current row is
  // processed -- a wave takes its next row index one row early -- so that its ~1 us latency hides behind the pair loop
  // (unpipelined, that load was most of the count time: 44 rows per wave x 1.5 us per vertex).
  auto grab = [&]() {
    int i = 0;
    if (lane == 0) i = atomicAdd(next_row, 1);
    return readfirst(i);
  };
  auto load_row = [&](const int i) -> unsigned {
    if (WHOLE) return (lane < stride) ? bits[i * ps + lane] : 0u;
    return gm[(size_t)min(i, d - 1) * stride + min(lane, stride - 1)];  // unconditional, clamped; masked by the caller
  };
  int inext = 0;
  unsigned mnext = 0u;
  if (!WHOLE) {
    inext = grab();
    mnext = (inext < jend) ? load_row(inext) : 0u;
  }
  for (;;) {
    int i;
    unsigned mi;
    if (WHOLE) {
      i = grab();
      if (i >= jend) break;
      mi = load_row(i);
    } else {
      i = inext;
      if (i >= jend) break;
      mi = (lane < stride) ? mnext : 0u;
      inext = grab();
      mnext = (inext < jend) ? load_row(inext) : 0u;  // wave-uniform condition
    }
    // nothing of row i inside this block's columns: no pair of it counts here
    if (__ballot(mi != 0u && lane >= c0 && lane < c0 + cwb) == 0ull) continue;
    // row i's words of this block, in registers (every lane reads the same addresses: LDS broadcast)
    uint4 mr[NQ];
    const uint4 *ri = reinterpret_cast<const uint4 *>(&bits[i * ps]);
#pragma unroll
    for (int k = 0; k < NQ; ++k) mr[k] = ri[k];
    int n = 0, start = 0;  // wave-uniform: list entries [start, n) are queued
    auto tile = [&](const int cnt) {
      const int idx = start + lane;
      const int j = (int)plist[min(idx, start + cnt - 1)];
      const uint4 *rj = reinterpret_cast<const uint4 *>(&bits[j * ps]);
      // first useful unit of the tile (wave-uniform): the list is ascending, lane 0 holds the smallest j
      // (rows of <= 5 units do not pay for the dispatch: measured, class S 7.5 -> 9.0 ms with it, profiles/r03/ab_clique4_steps.txt)
      const int kmin = (NQ >= 6 && topo) ? min(max(((readfirst(j) >> 5) - c0) >> 2, 0), NQ - 1) : 0;
      unsigned a = 0;
      switch (kmin) {
#define GM_PAIR_CASE(K) \
  case K: if constexpr (K < NQ) a = pair_popc<NQ, K>(mr, rj); break;
        GM_PAIR_CASE(0) GM_PAIR_CASE(1) GM_PAIR_CASE(2) GM_PAIR_CASE(3) GM_PAIR_CASE(4) GM_PAIR_CASE(5) GM_PAIR_CASE(6) GM_PAIR_CASE(7)
        GM_PAIR_CASE(8)
#undef GM_PAIR_CASE
        default: break;
      }
      c += (lane < cnt) ? a : 0u;
    };
    const int cbeg = topo ? ((i + 1) & ~63) : 0;
    for (int cc = cbeg; cc < jend; cc += 64) {  // (cc is a multiple of 64: word cc / 32 is even, at most 62)
      const unsigned w0 = (unsigned)readlane((int)mi, cc >> 5), w1 = (unsigned)readlane((int)mi, (cc >> 5) + 1);
      const unsigned wsel = (lane < 32) ? w0 : w1;
      const unsigned long long m = __ballot((((wsel >> (lane & 31)) & 1u) != 0u) & (cc + lane < jend));
      if (m == 0ull) continue;
      if (__builtin_amdgcn_inverse_ballot_w64(m)) plist[n + rank_below(m)] = (unsigned short)(cc + lane);
      n += __popcll(m);
      if (n - start >= GM_WAVE) {
        wave_sync();  // the list entries written above are read below
        // (two tiles of pairs in flight per wave -- pair_popc over two rows j at once -- was built: 128 - 140 VGPRs with spills against 100,
        // the scheduler hoists every read to the front; not kept)
        do {
          tile(GM_WAVE);
          start += GM_WAVE;
        } while (n - start >= GM_WAVE);
      }
      if (n + GM_WAVE > LISTCAP) {  // the next step may not fit: move the < 64 queued entries to the front
        const int rest = n - start;
        wave_sync();
        const unsigned short v = plist[min(start + min(lane, max(rest - 1, 0)), LISTCAP - 1)];
        wave_sync();
        if (lane < rest) plist[lane] = v;
        n = rest;
        start = 0;
      }
    }
    if (n > start) {
      wave_sync();
      tile(n - start);
    }
    wave_sync();  // the list is rewritten by the next row
    if (c > 0x7fffffffu) { tot += (unsigned long long)c; c = 0; }
  }
  return tot + (unsigned long long)c;
}

template <int WAVES, int WORDS, int LISTCAP, bool WHOLE>
__global__ __launch_bounds__(WAVES *GM_WAVE, 1) void clique_count_kernel(const CliqueCountParams p) {
  __shared__ CountLds<WAVES, WORDS, LISTCAP> S;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1), wave = tid >> 6;
  unsigned short *plist = S.plist[wave];
  unsigned long long tot = 0, t_load = 0, t_count = 0;
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = S.queue_pos;
    // class X: a queue entry is ONE COLUMN BLOCK of a vertex (entry = vertex * 8 + block; a vertex has at most 8) -- the blocks are
    // independent sums, and a rank's share of the few X vertices (R-MAT-22 ef 28: 5.4 K in all, 2.6 per CU and rank of eight) balances
    // block by block instead of waiting for whole vertices
    constexpr int kBlkShift = WHOLE ? 0 : 3;
    if (q >= ((unsigned)p.count << kBlkShift)) break;
    const int slot = p.slots[q >> kBlkShift];
    const int only_block = WHOLE ? -1 : (int)(q & 7u);
    const int u = p.verts[slot];
    const int d = p.rp[u + 1] - p.rp[u], stride = (d + 31) >> 5;
    const unsigned *__restrict__ gm = p.mat + p.base[slot];
    // column blocks: the fewest equal blocks of cw words whose padded copy (d rows of ps words) fits the LDS budget and the
    // register budget of a row (one block for classes S / L). d <= kWideMaxDeg: a block of <= 8 words always fits.
    int cw = stride, ps = clique_copy_stride(d, cw, WORDS);
    for (int nb = 2; (long long)d * ps > WORDS || ps > 4 * kCountMaxQ; ++nb) {
      cw = (stride + nb - 1) / nb;
      ps = clique_copy_stride(d, cw, WORDS);
    }
    const int nq = ps >> 2;
    for (int c0 = 0; c0 < stride; c0 += cw) {
      if (only_block >= 0 && c0 != only_block * cw) continue;  // (workgroup-uniform)
      const unsigned long long t0 = p.profile ? wall_clock64() : 0ull;
      const int cwb = min(cw, stride - c0);  // words of this block (the last one may be narrower; pads are zero)
      __syncthreads();                       // the previous block / vertex is no longer read
      // copy the block: a wave moves four rows per trip (four independent coalesced reads of <= 144 B in flight); pads zeroed
      // (triangular matrices: the rows at or beyond the block's last column are neither counted nor looked up -- count_block)
      const int drows = p.topo ? min(d, (c0 + cwb) * 32) : d;
      for (int row0 = wave * 4; row0 < drows; row0 += WAVES * 4) {
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gm[(size_t)min(row0 + k, d - 1) * stride + c0 + min(lane, cwb - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (row0 + k < drows && lane < ps) S.bits[(row0 + k) * ps + lane] = (lane < cwb) ? v[k] : 0u;
      }
      if (tid == 0) S.next_row = 0;
      __syncthreads();
      const unsigned long long t1 = p.profile ? wall_clock64() : 0ull;
      // (WHOLE instantiations -- classes S / L -- only ever see one block: clique_count_class and the loop above agree)
#define GM_COUNT_CASE(NQ) \
  case NQ: tot += count_block<NQ, WHOLE, LISTCAP>(S.bits, plist, &S.next_row, gm, d, stride, lane, c0, cwb, p.topo != 0); break;
      switch (nq) {
        GM_COUNT_CASE(1)
        GM_COUNT_CASE(2)
        GM_COUNT_CASE(3)
        GM_COUNT_CASE(4)
        GM_COUNT_CASE(5)
        GM_COUNT_CASE(6)
        GM_COUNT_CASE(7)
        GM_COUNT_CASE(8)
        default: tot += count_block<9, WHOLE, LISTCAP>(S.bits, plist, &S.next_row, gm, d, stride, lane, c0, cwb, p.topo != 0); break;
      }
#undef GM_COUNT_CASE
      if (p.profile) {
        const unsigned long long t2 = wall_clock64();
        t_load += t1 - t0;
        t_count += t2 - t1;
      }
    }
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64(tot);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
  if (p.profile && tid == 0) {
    atomicAdd(&p.profile[0], t_load);
    atomicAdd(&p.profile[1], t_count);
    atomicAdd(&p.profile[3], 1ull);
  }
}

// class S: 4 waves, 32 KB of bits, d+ < 512 -> 37 KB, four workgroups per CU
// class L: 16 waves, 112 KB of bits, whole matrices up to d+ = 896 -> 144 KB, one workgroup per CU
// class X: 16 waves, 112 KB of bits, column blocks of rows up to d+ = 2048 -> 144 KB, one workgroup per CU
using CountLdsS = CountLds<kCountWavesS, kCountWordsS, 512>;
using CountLdsL = CountLds<kCountWavesL, kCountWordsL, 1024>;
using CountLdsX = CountLds<kCountWavesX, kCountWordsL, 1024>;

size_t clique_count_lds_bytes(int cls) { return cls == 0 ? sizeof(CountLdsS) : cls == 1 ? sizeof(CountLdsL) : sizeof(CountLdsX); }
int clique_count_threads(int cls) { return GM_WAVE * (cls == 0 ? kCountWavesS : cls == 1 ? kCountWavesL : kCountWavesX); }

hipError_t launch_clique_count(int cls, const CliqueCountParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(CountLdsL) <= 163840 && sizeof(CountLdsX) <= 163840, "classes L / X must fit the 160 KB of one CU");
  static_assert((long long)kWideMaxDeg * 8 <= kCountWordsL, "column blocks of 5..8 words must fit for the widest row");
  const dim3 grid((unsigned)grid_blocks);
  if (cls == 0) hipLaunchKernelGGL((clique_count_kernel<kCountWavesS, kCountWordsS, 512, true>), grid, dim3(kCountWavesS * GM_WAVE), 0, stream, p);
  else if (cls == 1) hipLaunchKernelGGL((clique_count_kernel<kCountWavesL, kCountWordsL, 1024, true>), grid, dim3(kCountWavesL * GM_WAVE), 0, stream, p);
  else hipLaunchKernelGGL((clique_count_kernel<kCountWavesX, kCountWordsL, 1024, false>), grid, dim3(kCountWavesX * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_wide_kernel() {}
void gm_touch_wide() { hipLaunchKernelGGL(gm_touch_wide_kernel, dim3(1), dim3(1), 0, 0); }

