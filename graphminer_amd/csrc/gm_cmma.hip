// gm_cmma.hip -- k-clique (k = 4), the second DFS level of the WIDE vertices on the MATRIX CORES.
//
// For a vertex u with d = d+(u) the first level (gm_cbuild.hip) leaves the d x d adjacency bit-matrix M of N+(u) in the arena, and the
// number of 4-cliques with smallest member u is
//     sum_i sum_{j in M_i} popc(M_i & M_j)  =  sum_{i,j} M_ij * (M M^T)_ij                 (clique4_warp_edge.cuh:19-27 on bit rows)
// -- a masked binary matrix product.  The round-3 kernels (gm_wide.hip) evaluate the left form on the vector ALU: one v_and + v_bcnt
// per 32 bit-products and lane, 30 ms for the com-Orkut stand-in whose matrices are 35 % dense above the diagonal.  Here the right
// form runs as FP4 MFMA: a set bit is the E2M1 value 1.0 (nibble 0010), v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales multiplies
// 32 x 64 by 64 x 32 of them in 32 cycles (65,536 bit-products per instruction against 2,048 per v_and), the f32 accumulators hold
// exact integers <= 2048.  "No MFMA" in the north star describes the sorted-list intersections; this half of the pattern IS a GEMM.
//   * operands: lane l of a fragment holds row (l & 31), columns 32 * (l >> 5) .. + 31 of a 32 x 64 bit tile = ONE 32-bit word of the
//     matrix, expanded in registers to 32 nibbles with 5 - 7 VALU (shifts and masks: nibble p of register r is bit 4 p + r; mma_expand_a /
//     _b below).  The order of the columns inside a fragment does not matter -- both operands are rows of the same matrix, laid out
//     the same way -- so no transposition is ever needed (M M^T: both operands are "row-major in k");
//   * a wave owns a 64 x 64 block of (i, j) pairs = 2 x 2 accumulator tiles and runs over the column steps of 64; with a topological
//     numbering M is strictly upper triangular, so only blocks I <= J and steps K >= J exist (1/6 of the cube), and a block whose
//     64 x 64 mask is empty is skipped;
//   * the mask: the product is taken as C'[j][i] (rows of the J block as the A operand), so a lane holds column i = l & 31 of C' and
//     its 16 accumulator registers are j = (r & 3) + 8 (r >> 2) + 4 (l >> 5) -- bits of ONE word M[i][32 J ..] it loads once per tile;
//   * the matrix (or a column block of it, rows of > 1024 entries) sits in LDS with a row stride = 2 mod 4 words: the 64 lanes of a
//     fragment read 64 different banks.
#include "gm_flat.h"

namespace gm {

typedef int mma_v8i __attribute__((ext_vector_type(8)));
typedef float mma_v16f __attribute__((ext_vector_type(16)));

template <int WORDS>
struct alignas(16) MmaLds {
  unsigned bits[WORDS];  // the (column block of the) matrix: rows of `ps` words, rows and columns beyond the matrix are zero
  int next_task;
  unsigned queue_pos;
  int pad_[2];
};

// (round 5: the two operands encode a set bit differently -- bit class r = 0 / 1 / 2 / 3 is 0.5 / 1 / 2 / 2 in the A operand and 2 / 1 / 0.5 /
// 0.5 in the B operand, every product of two set bits exactly 1 -- so that five and seven vector instructions expand a word instead of seven
// and seven: gm_ctc.hip ctc_expand_a / _b)
__device__ __forceinline__ mma_v8i mma_expand_a(const unsigned x) {
  mma_v8i r = {(int)(x & 0x11111111u), (int)(x & 0x22222222u), (int)(x & 0x44444444u), (int)((x >> 1) & 0x44444444u), 0, 0, 0, 0};
  return r;
}
__device__ __forceinline__ mma_v8i mma_expand_b(const unsigned x) {
  mma_v8i r = {(int)((x << 2) & 0x44444444u), (int)(x & 0x22222222u), (int)((x >> 2) & 0x11111111u), (int)((x >> 3) & 0x11111111u), 0, 0, 0, 0};
  return r;
}

// sum of the accumulators whose mask bit is set: register r of this lane is bit (r & 3) + 8 (r >> 2) of w (already shifted by 4 (l >> 5))
// (round 6: two registers per v_pk_fma_f32 -- 24 instead of 32 + 8 instructions per tile -- made the whole pattern 0.7 ms SLOWER: not kept)
__device__ __forceinline__ float mma_masked_sum(const mma_v16f &acc, const unsigned w) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned mb = (w >> q) & 0x01010101u;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) s = fmaf(acc[q + 4 * pp], (float)((mb >> (8 * pp)) & 0xffu), s);
  }
  return s;
}

// task t of a triangular matrix = the block pair (JB, IB), IB <= JB, t = JB (JB + 1) / 2 + IB: a table instead of a square root and two correcting
// loops per block (round 6: a block of a d+ <= 512 matrix is ~10 MFMA, and everything around them counts)
constexpr int kMmaTriMax = 33 * 32 / 2;  // nJ <= 32 (2048 rows)
struct MmaTri {
  unsigned short v[kMmaTriMax];
  constexpr MmaTri() : v() {
    int t = 0;
    for (int jb = 0; jb < 32; ++jb)
      for (int ib = 0; ib <= jb; ++ib) v[t++] = (unsigned short)((jb << 8) | ib);
  }
};
__constant__ MmaTri kMmaTri = MmaTri();

template <int WAVES, int WORDS, bool BLOCKS>
__global__ __launch_bounds__(WAVES *GM_WAVE) void clique_mma_kernel(const CliqueCountParams p) {
  __shared__ MmaLds<WORDS> S;
  constexpr int NT = WAVES * GM_WAVE;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1);
  const int l31 = lane & 31, h = lane >> 5;
  const bool topo = p.topo != 0;
  unsigned long long tot = 0;
  // (round 6: the dequeue of the NEXT entry is issued before the current one is worked on, and an entry's {d+, matrix offset} is one record
  // in queue order -- the chain atomic -> slot -> vertex -> row bounds / offset -> matrix words was five dependent round trips per vertex)
  if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
  __syncthreads();
  for (;;) {
    const unsigned q = S.queue_pos;
    // column blocks: a queue entry is ONE block of a vertex (entry = slot * 8 + block) -- independent sums, and a rank's share of the
    // few widest vertices balances block by block
    constexpr int kBlkShift = BLOCKS ? 3 : 0;
    if (q >= ((unsigned)p.count << kBlkShift)) break;
    __syncthreads();  // (everybody has read the queue word)
    unsigned qn = 0u;
    if (tid == 0) qn = atomicAdd(p.queue, 1u);  // in flight while this entry is counted; published at the bottom
    const int4 qr = p.qrec[q >> kBlkShift];
    const int d = qr.x, stride = (d + 31) >> 5;
    const unsigned *__restrict__ gm = p.mat + (((unsigned long long)(unsigned)qr.z << 32) | (unsigned long long)(unsigned)qr.y);
    const int cw = BLOCKS ? clique_mma_block_words(d, WORDS) : ((stride + 1) & ~1);
    const int c0 = BLOCKS ? (int)(q & 7u) * cw : 0;
    if (c0 >= stride) {  // (workgroup-uniform: this vertex has fewer blocks)
      if (tid == 0) S.queue_pos = qn;
      __syncthreads();
      continue;
    }
    const int cwb = min(cw, stride - c0);
    const int ps = clique_mma_stride(cw);
    // triangular matrices: rows at or beyond the block's last column have no bit in it, and no pair (i, j) with j there counts
    const int drows = topo ? min(d, (c0 + cwb) * 32) : d;
    const int rows_alloc = (drows + 63) & ~63;
    {  // copy: LDS word idx = row * ps + col; pads and the rows beyond the matrix are zeroed
      const int total = rows_alloc * ps;
      const unsigned magic = 0xffffffffu / (unsigned)ps + 1u;  // idx / ps for idx < 2^26
      for (int i0 = tid; i0 < total; i0 += 4 * NT) {
        unsigned v[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = i0 + k * NT;
          const int row = (int)__umulhi((unsigned)idx, magic), col = idx - row * ps;
          in[k] = idx < total && row < drows && col < cwb;
          v[k] = gm[in[k] ? (size_t)row * stride + c0 + col : 0];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i0 + k * NT < total) S.bits[i0 + k * NT] = in[k] ? v[k] : 0u;
      }
    }
    if (tid == 0) S.next_task = 0;
    __syncthreads();
    const int nJ = rows_alloc >> 6;
    const int ntasks = topo ? nJ * (nJ + 1) / 2 : nJ * nJ;
    const int kb0 = c0 >> 1, kb1 = (c0 + cwb + 1) >> 1;  // column steps of 64 (c0 is even)
    unsigned c = 0;
    for (;;) {
      int t = 0;
      if (lane == 0) t = atomicAdd(&S.next_task, 1);
      t = readfirst(t);
      if (t >= ntasks) break;
      int JB, IB;
      if (topo) {  // t = JB (JB + 1) / 2 + IB, IB <= JB: the blocks with the longest column range first
        const unsigned e = kMmaTri.v[t];  // (t is wave-uniform: a scalar load)
        JB = (int)(e >> 8);
        IB = (int)(e & 255u);
      } else {
        JB = t / nJ;
        IB = t - JB * nJ;
      }
      const int ks0 = topo ? max(kb0, JB) : kb0;
      if (ks0 >= kb1) continue;
      // the mask of the block: word (32 J-tile) of row i = the lane's row of the I tile (from the arena: L2)
      unsigned mw[2][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int row = IB * 64 + ii * 32 + l31;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int wc = JB * 2 + jj;
          if (!BLOCKS) {  // the whole matrix is in LDS (rows and columns beyond it are zero there: rows_alloc x ps words)
            mw[jj][ii] = wc < ps ? S.bits[row * ps + wc] : 0u;
          } else {  // (a column block: the mask's column may lie outside it -- from the arena, L2)
            const bool in = row < d && wc < stride;
            const unsigned v = gm[in ? (size_t)row * stride + wc : 0];
            mw[jj][ii] = in ? v : 0u;
          }
        }
      }
      if (__ballot((mw[0][0] | mw[0][1] | mw[1][0] | mw[1][1]) != 0u) == 0ull) continue;  // no pair (i, j) in this block
      mma_v16f acc[2][2];
      const int xi = (IB * 64 + l31) * ps + h - c0, xj = (JB * 64 + l31) * ps + h - c0, hop = 32 * ps;
      {  // the first column step takes a ZERO C operand (an inline constant of the MFMA) instead of 64 registers cleared per block
        const mma_v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int ks = ks0;
        const unsigned wi0 = S.bits[xi + 2 * ks], wi1 = S.bits[xi + 2 * ks + hop], wj0 = S.bits[xj + 2 * ks], wj1 = S.bits[xj + 2 * ks + hop];
        const mma_v8i fi0 = mma_expand_b(wi0), fi1 = mma_expand_b(wi1), fj0 = mma_expand_a(wj0), fj1 = mma_expand_a(wj1);
        acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj0, fi0, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj0, fi1, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj1, fi0, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj1, fi1, zero, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      for (int ks = ks0 + 1; ks < kb1; ++ks) {
        const unsigned wi0 = S.bits[xi + 2 * ks], wi1 = S.bits[xi + 2 * ks + hop], wj0 = S.bits[xj + 2 * ks], wj1 = S.bits[xj + 2 * ks + hop];
        const mma_v8i fi0 = mma_expand_b(wi0), fi1 = mma_expand_b(wi1), fj0 = mma_expand_a(wj0), fj1 = mma_expand_a(wj1);
        // unit scales (E8M0 127); formats: 4 = FP4 (E2M1) for both operands
        acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj0, fi0, acc[0][0], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj0, fi1, acc[0][1], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj1, fi0, acc[1][0], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fj1, fi1, acc[1][1], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      float s = 0.f;  // <= 4 tiles x 16 registers x 2048: exact in f32
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) s += mma_masked_sum(acc[jj][ii], mw[jj][ii] >> (4 * h));
      c += (unsigned)s;
      if (c > 0x7fffffffu) {
        tot += (unsigned long long)c;
        c = 0;
      }
    }
    tot += (unsigned long long)c;
    if (tid == 0) S.queue_pos = qn;
    __syncthreads();  // the block is rewritten by the next queue entry
  }
  const unsigned long long s0 = wave_sum_u64(tot);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

using MmaLdsS = MmaLds<kMmaWordsS>;
using MmaLdsL = MmaLds<kMmaWordsL>;

size_t clique_mma_lds_bytes(int cls) { return cls == 0 ? sizeof(MmaLdsS) : sizeof(MmaLdsL); }
int clique_mma_threads(int cls) { return GM_WAVE * (cls == 0 ? kMmaWavesS : kMmaWavesL); }

hipError_t launch_clique_mma(int cls, const CliqueCountParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(sizeof(MmaLdsL) <= 163840, "the big instantiation must fit the 160 KB of one CU");
  static_assert(sizeof(MmaLdsS) * 4 <= 163840, "four small workgroups per CU");
  const dim3 grid((unsigned)grid_blocks);
  if (cls == 0) hipLaunchKernelGGL((clique_mma_kernel<kMmaWavesS, kMmaWordsS, false>), grid, dim3(kMmaWavesS * GM_WAVE), 0, stream, p);
  else if (cls == 1) hipLaunchKernelGGL((clique_mma_kernel<kMmaWavesL, kMmaWordsL, false>), grid, dim3(kMmaWavesL * GM_WAVE), 0, stream, p);
  else hipLaunchKernelGGL((clique_mma_kernel<kMmaWavesL, kMmaWordsL, true>), grid, dim3(kMmaWavesL * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_cmma_kernel() {}
void gm_touch_cmma() { hipLaunchKernelGGL(gm_touch_cmma_kernel, dim3(1), dim3(1), 0, 0); }
