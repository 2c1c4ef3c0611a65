// gm_wide.hip -- k-clique, phase 2 for WIDE vertices: the second DFS level counted from a big-LDS copy of the vertex's
// adjacency bit-matrix (see "k-clique, wide vertices" in gm_mine.h; phase 1 = clique_build_kernel below).
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

// ---- phase 1: build rows of a wide vertex ---------------------------------------------------------------------------------
// The chunk is a ROW RANGE of a wide vertex: the task edges (u, A[t0 .. t0 + rows)), A = N+(u), rows <= kBuildRowsPerChunk.
// The workgroup stages A and its hashed filter once; then every WAVE takes batches of kBuildBatchRows edges on its own, builds
// their rows in its private 1 KB of LDS exactly like a staged chunk of the mining kernel (pass X: stream N+(v) through the
// filter, bisect the survivors in LDS; pass Y: keys of A bisect N+(v) in HBM / L2) and writes the finished rows straight into
// the vertex's matrix in the arena -- rows are independent, so there is no workgroup barrier between batches. (The first
// version built 32-row groups with a barrier per group: the group waited for its slowest wave, and the batch size decided
// the kernel: 32-edge batches 139 ms, 8: 69.6, 4: 64 -- profiles/r02/ab_clique4_build_batch.log.) Building is latency-bound
// streaming -- occupancy is what counts -- so the workgroup is lean: 4 KB stage + 4 KB filter + 4 x 1 KB of rows + 4 x 2.4 KB of
// pass scratch = 22 KB, seven workgroups (28 waves) per CU like TC; mine_kernel<PAT_CLIQUE4> (29.5 KB, 95 VGPRs) runs five.
// Rows longer than the 1024-entry stage (d+ 1025..2048) are searched in HBM / L2.
struct alignas(16) BuildLds {
  int stage[kStageCapClique];     // N+(u) (first member: the bisection may read past the row, never past LDS)
  unsigned fbits[kFilterWords];   // hashed membership filter of the row (salt 0)
  unsigned bits[kWavesPerBlock][kBuildBatchRows * (kWideMaxDeg / 32)];  // per wave: the rows of its current batch
  int next_batch;
  unsigned queue_pos;
  WaveLdsLean w[kWavesPerBlock];
};

__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE, 7) void clique_build_kernel(const CliqueBuildParams p) {
  __shared__ BuildLds B;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int2 *__restrict__ edesc = p.g.edesc;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1), wave = tid >> 6;
  constexpr int NT = kWavesPerBlock * GM_WAVE;
  WaveLdsLean &L = B.w[wave];
  unsigned *wb = B.bits[wave];
  for (;;) {
    if (tid == 0) B.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const ChunkRec r = p.chunks[q];
    const int u = r.u_begin, ru = rp[u], d = rp[u + 1] - ru, stride = (d + 31) >> 5;
    const int t0 = r.e_begin - ru, rows = r.e_end - r.e_begin;
    const bool staged = d <= kStageCapClique;
    if (staged) {
      for (int i = tid; i < d; i += NT) B.stage[i] = col[ru + i];
      for (int i = tid; i < kFilterWords; i += NT) B.fbits[i] = 0u;
    }
    if (tid == 0) B.next_batch = 0;
    __syncthreads();
    if (staged) {
      for (int i = tid; i < d; i += NT) {
        const unsigned h = filter_hash(B.stage[i], 0u);
        atomicOr(&B.fbits[h >> 5], 1u << (h & 31u));
      }
      __syncthreads();
    }
    unsigned *__restrict__ gm = p.mat + p.base[r.pad_ - 1] + (size_t)t0 * stride;  // the chunk's rows in the vertex's matrix
    constexpr int bsz = kBuildBatchRows;
    for (;;) {  // waves take batches on their own: no workgroup barrier until the chunk is done
      int bi = 0;
      if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
      bi = readfirst(bi);
      const int l0 = bi * bsz;  // first local row of the batch
      if (l0 >= rows) break;
      const int nr = min(bsz, rows - l0);
      for (int i = lane; i < nr * stride; i += GM_WAVE) wb[i] = 0u;
      const int lr = l0 + lane;
      const bool valid = (lane < nr);
      int rv = 0, b = 0;
      if (valid) {
        if (edesc) {
          const int2 de = edesc[r.e_begin + lr];
          rv = de.x;
          b = de.y;
        } else {
          const int v = col[r.e_begin + lr];
          rv = rp[v];
          b = rp[v + 1] - rv;
        }
      }
      const bool act = valid && b > 0;
      bool dirx = false;
      if (act) {
        if (staged) {  // the direction rule of the mining kernel (process_chunk)
          const float cx = (float)b * (float)(p.cost_x_base + p.cost_x_step * bitlen(d));
          const float cy = (float)d * (float)(p.cost_y_base + p.cost_y_step * bitlen(b));
          dirx = cx <= cy;
        } else {
          dirx = b <= d;
        }
      }
      wave_sync();  // the row buffer is zeroed
      auto set_bit = [&](const int owner, const int cbit) { atomicOr(&wb[owner * stride + (cbit >> 5)], 1u << (cbit & 31)); };
      auto actx = [&](bool f, int owner, int, int pos, int, int) { if (f) set_bit(owner, pos); };   // pos: position in N+(u)
      auto acty = [&](bool f, int owner, int kidx, int, int, int) { if (f) set_bit(owner, kidx); };  // kidx: index of the key in N+(u)
      if (staged) flat_pass_filtered(L, B.stage, B.fbits, col, lane, (act && dirx) ? b : 0, rv, 0, d, 0, actx);
      else flat_pass<SEARCH_HBM>(L, B.stage, col, nullptr, lane, (act && dirx) ? b : 0, rv, ru, d, actx);
      flat_pass<SEARCH_HBM>(L, B.stage, col, nullptr, lane, (act && !dirx) ? d : 0, ru, rv, b, acty);
      wave_sync();
      for (int i = lane; i < nr * stride; i += GM_WAVE) gm[(size_t)l0 * stride + i] = wb[i];  // finished rows: contiguous in the arena
      wave_sync();
    }
    __syncthreads();  // every wave is done with the stage / filter
  }
}

size_t clique_build_lds_bytes() { return sizeof(BuildLds); }

hipError_t launch_clique_build(const CliqueBuildParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(BuildLds) * 7 <= 163840, "seven build workgroups per CU");
  hipLaunchKernelGGL(clique_build_kernel, dim3((unsigned)grid_blocks), dim3(kWavesPerBlock * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

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
template <int NQ, bool WHOLE>
__device__ __forceinline__ unsigned long long count_block(const unsigned *__restrict__ bits, unsigned short *__restrict__ plist, int *next_row,
                                                          const unsigned *__restrict__ gm, const int d, const int stride, const int lane) {
  constexpr int ps = 4 * NQ;
  unsigned long long tot = 0;
  unsigned c = 0;
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
    mnext = (inext < d) ? load_row(inext) : 0u;
  }
  for (;;) {
    int i;
    unsigned mi;
    if (WHOLE) {
      i = grab();
      if (i >= d) break;
      mi = load_row(i);
    } else {
      i = inext;
      if (i >= d) break;
      mi = (lane < stride) ? mnext : 0u;
      inext = grab();
      mnext = (inext < d) ? load_row(inext) : 0u;  // wave-uniform condition
    }
    // row i's words of this block, in registers (every lane reads the same addresses: LDS broadcast)
    uint4 mr[NQ];
    const uint4 *ri = reinterpret_cast<const uint4 *>(&bits[i * ps]);
#pragma unroll
    for (int k = 0; k < NQ; ++k) mr[k] = ri[k];
    // the j's of row i, 1024 columns (32 words) at a time: the position list of a wave holds 1024 entries (2 KB)
    for (int h0 = 0; h0 < stride; h0 += 32) {
      const int cwn = (lane >= h0 && lane < h0 + 32) ? __popc(mi) : 0;
      const int incl = wave_incl_scan_add(cwn);
      const int total = readlane(incl, GM_WAVE - 1);
      if (total == 0) continue;
      wave_sync();  // the previous half's list is no longer read
      if (cwn) {
        unsigned x = mi;
        int k = incl - cwn;
        while (x) {
          plist[k++] = (unsigned short)(lane * 32 + (__ffs((int)x) - 1));
          x &= x - 1;
        }
      }
      wave_sync();
#pragma unroll 1
      for (int t = 0; t < total; t += GM_WAVE) {  // (not unrolled: one step already has up to 9 independent 16-byte reads in flight)
        const int idx = t + lane;
        const int j = (int)plist[min(idx, total - 1)];
        const uint4 *rj = reinterpret_cast<const uint4 *>(&bits[j * ps]);
        unsigned a = 0;
        constexpr int G = 5;  // units requested together: 9 = 5 + 4 keeps the kernel at 4 waves per SIMD without spills
#pragma unroll
        for (int k0 = 0; k0 < NQ; k0 += G) {
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
        c += (idx < total) ? a : 0u;
      }
    }
    wave_sync();
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
    if (q >= (unsigned)p.count) break;
    const int slot = p.slots[q];
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
      const unsigned long long t0 = p.profile ? wall_clock64() : 0ull;
      const int cwb = min(cw, stride - c0);  // words of this block (the last one may be narrower; pads are zero)
      __syncthreads();                       // the previous block / vertex is no longer read
      // copy the block: a wave moves four rows per trip (four independent coalesced reads of <= 144 B in flight); pads zeroed
      for (int row0 = wave * 4; row0 < d; row0 += WAVES * 4) {
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gm[(size_t)min(row0 + k, d - 1) * stride + c0 + min(lane, cwb - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (row0 + k < d && lane < ps) S.bits[(row0 + k) * ps + lane] = (lane < cwb) ? v[k] : 0u;
      }
      if (tid == 0) S.next_row = 0;
      __syncthreads();
      const unsigned long long t1 = p.profile ? wall_clock64() : 0ull;
      // (WHOLE instantiations -- classes S / L -- only ever see one block: clique_count_class and the loop above agree)
#define GM_COUNT_CASE(NQ) \
  case NQ: tot += count_block<NQ, WHOLE>(S.bits, plist, &S.next_row, gm, d, stride, lane); break;
      switch (nq) {
        GM_COUNT_CASE(1)
        GM_COUNT_CASE(2)
        GM_COUNT_CASE(3)
        GM_COUNT_CASE(4)
        GM_COUNT_CASE(5)
        GM_COUNT_CASE(6)
        GM_COUNT_CASE(7)
        GM_COUNT_CASE(8)
        default: tot += count_block<9, WHOLE>(S.bits, plist, &S.next_row, gm, d, stride, lane); break;
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
