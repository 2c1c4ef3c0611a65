// gm_cgather.hip -- k-clique (k = 4), first DFS level of the WIDE vertices GATHERED from a dense bitmap of the hub core.
//
// Row i of u's adjacency bit-matrix over N+(u) = [s_0 < s_1 < ...] is  M_u[i][j] = [s_i -> s_j is an edge]  (j > i under a topological
// numbering).  The streamed build (gm_cbuild.hip) finds those bits as a set intersection N+(u) ^ N+(s_i): a hashed lookup per streamed
// key, 44 VALU per 64 keys, 23 G keys on the com-Orkut stand-in -- 34.5 of the 42.5 ms left after the pair counts moved to the matrix
// cores.  But M_u is just the adjacency matrix of the subgraph induced by N+(u), and for a WIDE vertex (d+ > 256) of a DAG numbered by
// degree nearly all of N+(u) lies among the LAST few 10^4 ids -- the hubs.  Their adjacency is kept DENSE (gm_graph::d_core: core_h x
// core_h bits, one row per core vertex, built once per graph), and a row of M_u is a bit GATHER:
//     M_u[i][j] = core[s_i - base][s_j - base],   j > i, s_i >= base  (then every s_j > s_i is in the core too)
// -- lane j of a 64-column tile loads the word that holds its bit (the positions are ascending and dense at the top of the id range:
// 1.4 - 3.8 bytes of 64-byte lines per probe on R-MAT, each gathered dword serves ~3 probes), tests it, and the ballot IS 64 bits of
// the row: ~10 instructions per 64 probes, no hash, no table build, no LDS atomics.  96 % of the first-level work of the wide vertices
// of R-MAT graphs has its first endpoint in a core of nv / 32 vertices; the rows whose first endpoint lies below the core (the first
// entries of the ascending list) stay tasks of the streamed build.  Rows are stored to the matrix arena like the streamed build's;
// the pair counts (gm_cmma.hip) read them from there.
// (reference shape: src/clique/gpu_kernels/clique4_warp_edge.cuh:19-21 re-intersects N+(v0) ^ N+(v1) from global memory per edge)
#include <algorithm>
#include <cstdlib>
#include "gm_flat.h"
#pragma clang diagnostic ignored "-Winline-asm"  // (M0 on a clobber list: see cg_tile)

namespace gm {

constexpr int kCgWaves = 8;
constexpr int kCgUnroll = 8;  // tiles of a row whose gathers are in flight together (beyond its head)
constexpr int kCgHead = 8;    // tiles of the NEXT row of the wave requested ahead

struct alignas(16) CGatherLds {
  int2 info[kCbMaxDeg + GM_WAVE];  // per column j: {byte offset of the word of bit s_j - core_base in a core row, s_j - core_base} (-1: below the core)
  unsigned queue_pos;
  int k0;
  int pad_[2];
};

// tile t of row i: 64 probes of the core row at the columns' positions -> the 64 bits of the row, left in lanes 2t / 2t + 1 of w_out
// (CHECKED: the tile that holds the diagonal -- only columns beyond i count -- or the end of the row)
template <bool CHECKED>
__device__ __forceinline__ void cg_tile(const unsigned w, const int2 inf, const int t, const int i, const int d, const int lane, unsigned &w_out) {
  bool bit = __builtin_amdgcn_ubfe(w, (unsigned)inf.y, 1u) != 0u;  // (v_bfe_u32 takes the low five bits of the offset)
  if (CHECKED) {
    const int j = t * GM_WAVE + lane;
    bit = bit && j > i && j < d;
  }
  const unsigned long long m = __ballot(bit);
  // (v_writelane_b32: no clang builtin in ROCm 7.2.  Value and lane select are both scalar; gfx950 allows ONE SGPR on the constant
  // bus, so the lane select goes through M0 -- nothing else in this kernel uses M0)
  // (the same wait states behind the ballot as cg_tile_k -- the hazard recogniser does not look inside inline assembly: VALU write of an
  // SGPR -> v_writelane reading it needs four; the s_mov counted as one of them and that was luck, not a rule.  VERDICT r5 weak 10)
  asm("s_nop 3\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(w_out) : "s"((unsigned)m), "s"(2 * t) : "m0");
  asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(w_out) : "s"((unsigned)(m >> 32)), "s"(2 * t + 1) : "m0");
}

// The same with the 64 bits left in lanes 2 K / 2 K + 1 of a GROUP register, K a compile-time constant: v_writelane_b32 takes the lane as an
// inline constant, no M0 -- the dynamic lane select cost two scalar instructions per v_writelane (s_add, s_mov m0: 37 scalar instructions
// per eight tiles, the one scalar unit of a CU 0.70 busy: profiles/r05/clique4_rmat22ef28_pmc_summary.txt).  cg_merge moves a group's sixteen
// lanes to lanes 2 t0 .. 2 t0 + 15 of the row register with one ds_bpermute_b32.
template <bool CHECKED, int K>
__device__ __forceinline__ void cg_tile_k(const unsigned w, const int2 inf, const int t, const int i, const int d, const int lane, unsigned &grp) {
  bool bit = __builtin_amdgcn_ubfe(w, (unsigned)inf.y, 1u) != 0u;
  if (CHECKED) {
    const int j = t * GM_WAVE + lane;
    bit = bit && j > i && j < d;
  }
  const unsigned long long m = __ballot(bit);
  // (the ballot is a VALU write of an SGPR pair: the compiler's hazard recogniser does not look inside inline assembly, and without the
  // wait states a v_writelane right behind the v_cmp now and then read the pair too early -- counts off by 0.3 % from run to run)
  // (all eight ballots of a group first, then ONE s_nop and the sixteen writes in one block: 26.1 -> 26.85 ms -- every tile then waits for
  // the group's last gather)
  asm("s_nop 3\n\tv_writelane_b32 %0, %1, %2" : "+v"(grp) : "s"((unsigned)m), "n"(2 * K));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(grp) : "s"((unsigned)(m >> 32)), "n"(2 * K + 1));
}
__device__ __forceinline__ void cg_merge(unsigned &w_out, const unsigned grp, const int t0, const int lane) {
  const int src = lane - 2 * t0;  // lane src of the group belongs to this lane of the row
  const unsigned v = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)grp);
  if ((unsigned)src < 2u * (unsigned)kCgUnroll) w_out = v;
}

template <int K, class HeadT>
__device__ __forceinline__ void cg_head(const HeadT &h, const int t0, const int ntiles, const int i, const int d, const int lane, unsigned &grp) {
  if constexpr (K < kCgHead) {
    if (t0 + K < ntiles) cg_tile_k<true, K>(h.w[K], h.inf[K], t0 + K, i, d, lane, grp);
    cg_head<K + 1>(h, t0, ntiles, i, d, lane, grp);
  }
}
template <int K>
__device__ __forceinline__ void cg_group(const unsigned (&w)[kCgUnroll], const int2 (&inf)[kCgUnroll], const int tb, const int i, const int d, const int lane,
                                         unsigned &grp) {
  if constexpr (K < kCgUnroll) {
    cg_tile_k<false, K>(w[K], inf[K], tb + K, i, d, lane, grp);
    cg_group<K + 1>(w, inf, tb, i, d, lane, grp);
  }
}
static_assert(kCgHead == kCgUnroll, "cg_merge moves sixteen lanes");

__global__ __launch_bounds__(kCgWaves *GM_WAVE) void cgather_kernel(const CGatherParams p) {
  __shared__ CGatherLds S;
  constexpr int NT = kCgWaves * GM_WAVE;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1), wave = readfirst(tid >> 6);  // (uniform: rows, tiles and row pointers live in SGPRs)
  // the core bitmap as ONE buffer resource: a gather is buffer_load_dword with the row's byte offset as the scalar offset and the word's
  // byte offset inside the row as the lane offset (global_load with a 64-bit lane address cost two more VALU per tile here)
  const __amdgpu_buffer_rsrc_t core = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(p.core), 0, (int)p.core_bytes, 0x00020000);
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = S.queue_pos;
    if (q >= (unsigned)p.count) break;
    const int slot = p.first_slot + (int)q;
    const int u = p.verts[slot];
    const int ru = p.rp[u], d = p.rp[u + 1] - ru, stride = (d + 31) >> 5;
    unsigned *__restrict__ mu = p.mat + p.base[slot];
    const int dpad = (d + GM_WAVE - 1) & ~(GM_WAVE - 1);
    for (int j = tid; j < dpad; j += NT) {
      const int pj = j < d ? p.col[ru + j] - p.core_base : -1;
      S.info[j] = make_int2(pj >= 0 ? (pj >> 5) << 2 : 0, pj);
    }
    if (tid == 0) S.k0 = lower_bound(p.col + ru, d, p.core_base);  // rows [0, k0): first endpoint below the core -- the streamed build's
    __syncthreads();
    const int k0 = S.k0;
    const int ntiles = dpad >> 6;
    // Two rows of a wave in flight: the gathers of the first kCgHead tiles of row i + kCgWaves are requested before the tiles of row i are
    // turned into bits -- a row is one dependent chain (column table -> buffer_load -> ballot -> writelane) and at the 32 waves a CU holds
    // the kernel waited 0.79 of its cycles for it (profiles/r04/clique4_rmat22ef28_pmc_summary.txt).
    struct Head {
      int2 inf[kCgHead];
      unsigned w[kCgHead];
    };
    auto issue = [&](const int i, Head &h) {  // (wave-uniform i; tiles at or beyond the row's last one are not requested)
      const int rowo = readfirst(S.info[i].y) * p.core_words * 4;
      const int t0 = (i + 1) >> 6;
#pragma unroll
      for (int k = 0; k < kCgHead; ++k) {
        if (t0 + k < ntiles) {
          h.inf[k] = S.info[(t0 + k) * GM_WAVE + lane];
          h.w[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, h.inf[k].x, rowo, 0);
        }
      }
    };
    auto finish = [&](const int i, const Head &h) {
      const int rowo = readfirst(S.info[i].y) * p.core_words * 4;
      unsigned w_out = 0u;  // lane L: word L of the row (the words below the diagonal are zero)
      const int t0 = (i + 1) >> 6;
      {
        unsigned grp = 0u;  // (tiles at or beyond the row's last one: zero words, never stored)
        cg_head<0>(h, t0, ntiles, i, d, lane, grp);
        cg_merge(w_out, grp, t0, lane);
      }
      int tb = t0 + kCgHead;
      for (; tb + kCgUnroll < ntiles; tb += kCgUnroll) {  // whole tiles beyond the head, kCgUnroll gathers in flight
        int2 inf[kCgUnroll];
        unsigned w[kCgUnroll];
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) inf[k] = S.info[(tb + k) * GM_WAVE + lane];
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) w[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf[k].x, rowo, 0);
        unsigned grp = 0u;
        cg_group<0>(w, inf, tb, i, d, lane, grp);
        cg_merge(w_out, grp, tb, lane);
      }
      for (; tb < ntiles; ++tb) {  // the rest, the row's last tile among them
        const int2 inf = S.info[tb * GM_WAVE + lane];
        cg_tile<true>((unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf.x, rowo, 0), inf, tb, i, d, lane, w_out);
      }
      if (lane < stride) mu[(size_t)i * stride + lane] = w_out;
    };
    int i = k0 + wave;
    Head cur, nxt;
    if (i < d) issue(i, cur);
    for (; i < d; i += kCgWaves) {
      const int in = i + kCgWaves;
      if (in < d) issue(in, nxt);
      finish(i, cur);
      cur = nxt;
    }
    __syncthreads();  // the column table is rewritten by the next vertex
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The BLOCKED gather (round 6; gm_mine.h CGatherBParams).  cgather_kernel above pulls, for every (vertex u, core neighbour s_i), the 128-byte
// lines of the 4 KB core row s_i that the columns N+(u) beyond s_i fall into -- ~12 probes per line at the bottom of the core, 53.6 GB for 21 GB
// of words (profiles/r05/clique4_rmat22ef28_pmc_summary.txt), at the rate of the fabric.  Here the loop nest is turned round: a BLOCK of
// consecutive core rows (only their words right of the diagonal: 16 rows at the bottom of the core, ~1000 at its top, where most vertices'
// rows are) sits in LDS, and the UNITS (vertex u, its rows inside the block: ~3 on R-MAT-22 ef 28) listed for the block once per plan are
// streamed past it.  A unit reads the part of N+(u) beyond its first row -- coalesced, 4 bytes per column, once for all its rows -- and every
// probe is one ds_read_b32; the bits of eight tiles are collected as above (ballot -> v_writelane into a group register) and the group's
// sixteen words are stored to the row directly.  A wave keeps THREE units in flight: the record, the row ids and the first eight column tiles of the unit
// after next are requested before the current one is probed (records: one vector load per sixteen units, extracted with v_readlane -- no scalar-load
// round trip per unit).
// Geometry (the A/B builds of profiles/r06/ab_clique4_blocked_gather.txt set these): the kernel is bound by the latency of each wave's own
// dependent chain (v_readlane -> v_add -> ds_read -> v_bfe -> v_cmp -> v_writelane), not by a unit or by bytes -- at four waves per SIMD (two
// 8-wave workgroups per CU, 109 registers, two units requested ahead, the next row's LDS reads issued early) it took 10.3 ms whatever the
// instruction count; EIGHT waves per SIMD (two 16-wave workgroups, 64 registers: one unit ahead, five table dwords ahead, no second set of
// word registers) 7.5 ms.
#ifndef CGB_WAVES
#define CGB_WAVES 16
#endif
#ifndef CGB_DEPTH
#define CGB_DEPTH 1
#endif
#ifndef CGB_WPE
#define CGB_WPE 8
#endif
#ifndef CGB_PINGPONG
#define CGB_PINGPONG 0
#endif
#ifndef CGB_HEAD
#define CGB_HEAD 5
#endif
#ifndef CGB_PAIR_MAX
#define CGB_PAIR_MAX 0
#endif
constexpr int kCgbWaves = CGB_WAVES;
struct alignas(16) CGatherBLds {
  unsigned img[kCgbWords];
  int rb4[kCgbMaxRows];  // per row of the block: byte offset of its image row, minus the bytes of its first stored word: + (q >> 5) * 4 = the word of column q
  unsigned queue_pos;
  int pad_[3];
};
struct CgbUnit {  // (wave-uniform)
  int d, i0, r;
  unsigned px, mbase;  // px: start of the vertex's column table
};
constexpr int kCgbHead = CGB_HEAD;  // dwords of a unit's table requested ahead: tile pairs T0 .. T0 + 8 (T0 = t0 >> 1) = the first two tile groups
struct CgbHead {
  unsigned rowq;           // lane l < min(r, 64): position q of row i0 + l
  unsigned hd[kCgbHead];
};
__device__ __forceinline__ unsigned long long cgb_mask_gt(const int i, const int lo) {  // lanes l of the tile at column lo with lo + l > i
  const int s = i + 1 - lo;  // first valid lane
  return s <= 0 ? ~0ull : (s >= 64 ? 0ull : (~0ull << s));
}
__device__ __forceinline__ unsigned long long cgb_mask_lt(const int d, const int lo) {  // ... with lo + l < d
  const int s = d - lo;
  return s >= 64 ? ~0ull : (s <= 0 ? 0ull : ((1ull << s) - 1ull));
}
__device__ __forceinline__ unsigned cgb_lds(const unsigned *img, const int byte_off) {
  return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(img) + byte_off);
}
// The ballots of N tiles -> lanes 0 .. 2 N - 1 of the group register (constant lanes: no M0), as ONE block of assembly: the wait states
// between the last v_cmp and the first v_writelane that reads its SGPR pair once (the hazard recogniser does not look inside inline assembly:
// cg_tile_k), and no s_nop between the v_writelane -- the compiler puts one between any two asm statements.
#define CGB_WL(K) "v_writelane_b32 %0, %" #K "\n\t"
template <int N>
__device__ __forceinline__ unsigned cgb_put(const unsigned long long (&m)[kCgUnroll]) {
  unsigned grp = 0u;
#define CGB_M(k) "s"(__builtin_amdgcn_readfirstlane((unsigned)m[k])), "s"(__builtin_amdgcn_readfirstlane((unsigned)(m[k] >> 32)))  /* (folds away on a scalar value; the "s" constraint alone does not force one) */
  if constexpr (N == 1)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1" : "+v"(grp) : CGB_M(0));
  else if constexpr (N == 2)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3" : "+v"(grp) : CGB_M(0), CGB_M(1));
  else if constexpr (N == 3)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2));
  else if constexpr (N == 4)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2), CGB_M(3));
  else if constexpr (N == 5)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2), CGB_M(3), CGB_M(4));
  else if constexpr (N == 6)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2), CGB_M(3), CGB_M(4), CGB_M(5));
  else if constexpr (N == 7)
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11\n\t"
                 "v_writelane_b32 %0, %13, 12\n\tv_writelane_b32 %0, %14, 13"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2), CGB_M(3), CGB_M(4), CGB_M(5), CGB_M(6));
  else
    asm volatile("s_nop 3\n\tv_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 8\n\tv_writelane_b32 %0, %10, 9\n\tv_writelane_b32 %0, %11, 10\n\tv_writelane_b32 %0, %12, 11\n\t"
                 "v_writelane_b32 %0, %13, 12\n\tv_writelane_b32 %0, %14, 13\n\tv_writelane_b32 %0, %15, 14\n\tv_writelane_b32 %0, %16, 15"
                 : "+v"(grp) : CGB_M(0), CGB_M(1), CGB_M(2), CGB_M(3), CGB_M(4), CGB_M(5), CGB_M(6), CGB_M(7));
#undef CGB_M
  return grp;
}
#undef CGB_WL
// the words of N tiles of one row -> their bits in the group register
template <int N>
__device__ __forceinline__ unsigned cgb_bits(const unsigned (&w)[kCgUnroll], const int (&inf)[kCgUnroll], const unsigned long long mfirst,
                                             const unsigned long long mlast) {
  unsigned long long m[kCgUnroll];
#pragma unroll
  for (int k = 0; k < kCgUnroll; ++k) m[k] = 0ull;
#pragma unroll
  for (int k = 0; k < N; ++k) m[k] = __ballot(__builtin_amdgcn_ubfe(w[k], (unsigned)inf[k], 1u) != 0u);  // (v_bfe_u32 takes the low five bits of the id)
  m[0] &= mfirst;
  m[N - 1] &= mlast;
  return cgb_put<N>(m);
}
template <int N>
__device__ __forceinline__ void cgb_read(const unsigned *img, const int rb4, const int (&wo)[kCgUnroll], unsigned (&w)[kCgUnroll]) {
#pragma unroll
  for (int k = 0; k < N; ++k) w[k] = cgb_lds(img, rb4 + wo[k]);
}
// The rows [g, gend) of a unit that share their diagonal tile t0, against ONE group of N column tiles (tb .. tb + N - 1): the LDS reads of the
// next row are requested before the current row's words are turned into bits, and the group's sixteen words go straight to the row in the
// arena -- a row's tile groups own disjoint words, so nothing is merged in registers and the COLUMNS of a group are taken once per unit, not
// once per row.  pr: the five table dwords that hold the group's tiles (tile kk = half (K0 + kk) & 1 of dword (K0 + kk) >> 1; K0 = tb & 1) --
// only the N tiles the group has are unpacked.  head (tb = t0): mgt = the lanes right of the first row's diagonal in tile t0 (one lane fewer
// per row), and the words below the diagonal's (zlane: lanes < 2 t0) are written as zeros.  mlt: the lanes of tile N - 1 that are columns of
// the vertex at all.
template <int N, int K0>
__device__ __forceinline__ void cgb_rows(const unsigned *img, const int rb4v, const unsigned (&pr)[5], int g, const int gend, const bool head,
                                         unsigned long long mgt, const unsigned long long mlt, const __amdgpu_buffer_rsrc_t mrs, int soff, const int stride4,
                                         const int voff_d, const int voff_z) {
  int inf[kCgUnroll], wo[kCgUnroll];
#pragma unroll
  for (int kk = 0; kk < kCgUnroll; ++kk) {
    inf[kk] = 0;
    wo[kk] = 0;
  }
#pragma unroll
  for (int kk = 0; kk < N; ++kk) {
    const int k = K0 + kk;
    inf[kk] = (k & 1) ? (int)(pr[k >> 1] >> 16) : (int)(pr[k >> 1] & 0xffffu);
    wo[kk] = (inf[kk] >> 3) & ~3;
  }
  const unsigned long long nh = head ? 0ull : ~0ull;  // (a group that is not the rows' first masks nothing in its first tile)
  unsigned wa[kCgUnroll], wb[kCgUnroll];
  // (the stores are buffer stores: the lanes that have no word of the row carry an offset beyond the resource and are dropped -- no EXEC juggling)
  auto row = [&](const unsigned (&w)[kCgUnroll]) {
    const unsigned grp = cgb_bits<N>(w, inf, mgt | nh, mlt);  // lanes 0 .. 15: the words wfirst .. wfirst + 15 of the row
    __builtin_amdgcn_raw_buffer_store_b32(grp, mrs, voff_d, soff, 0);
    if (head) __builtin_amdgcn_raw_buffer_store_b32(0u, mrs, voff_z, soff, 0);  // (wave-uniform) the words below the diagonal's
    soff += stride4;
    mgt <<= 1;
  };
#if CGB_PINGPONG
  cgb_read<N>(img, readlane(rb4v, g & (GM_WAVE - 1)), wo, wa);
  for (;;) {
    if (g + 1 < gend) cgb_read<N>(img, readlane(rb4v, (g + 1) & (GM_WAVE - 1)), wo, wb);  // (wave-uniform)
    row(wa);
    if (++g >= gend) break;
    if (g + 1 < gend) cgb_read<N>(img, readlane(rb4v, (g + 1) & (GM_WAVE - 1)), wo, wa);
    row(wb);
    if (++g >= gend) break;
  }
#else
  if constexpr (N <= CGB_PAIR_MAX) {  // two rows per trip: their LDS reads go out together and the loop's scalar bookkeeping is halved
    for (; g + 1 < gend; g += 2) {
      cgb_read<N>(img, readlane(rb4v, g & (GM_WAVE - 1)), wo, wa);
      cgb_read<N>(img, readlane(rb4v, (g + 1) & (GM_WAVE - 1)), wo, wb);
      row(wa);
      row(wb);
    }
  } else {
    (void)wb;
  }
  for (; g < gend; ++g) {
    cgb_read<N>(img, readlane(rb4v, g & (GM_WAVE - 1)), wo, wa);
    row(wa);
  }
#endif
}
template <int K0>
__device__ __forceinline__ void cgb_rows_n(const int nt, const unsigned *img, const int rb4v, const unsigned (&pr)[5], const int g, const int gend, const bool head,
                                           const unsigned long long mgt, const unsigned long long mlt, const __amdgpu_buffer_rsrc_t mrs, const int soff,
                                           const int stride4, const int voff_d, const int voff_z) {
  switch (nt) {  // (wave-uniform)
    case 1: cgb_rows<1, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 2: cgb_rows<2, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 3: cgb_rows<3, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 4: cgb_rows<4, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 5: cgb_rows<5, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 6: cgb_rows<6, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    case 7: cgb_rows<7, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
    default: cgb_rows<8, K0>(img, rb4v, pr, g, gend, head, mgt, mlt, mrs, soff, stride4, voff_d, voff_z); break;
  }
}

// (two workgroups of eight waves per CU -- the block images take 68 KB each -- are four waves per SIMD: 128 registers)
__global__ __launch_bounds__(kCgbWaves *GM_WAVE) __attribute__((amdgpu_waves_per_eu(CGB_WPE, CGB_WPE))) void cgatherb_kernel(const CGatherBParams p) {
  __shared__ CGatherBLds S;
  constexpr int NT = kCgbWaves * GM_WAVE;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1), wave = readfirst(tid >> 6);
  int cur_block = -1, q0 = 0;  // q0: position q of the block's first row
  for (;;) {
    if (tid == 0) S.queue_pos = atomicAdd(p.queue, 1u);
    __syncthreads();
    const unsigned q = S.queue_pos;
    if (q >= (unsigned)p.count) break;
    const int2 it = p.items[q];
    const int uend = p.items[q + 1].x;
    if (it.y != cur_block) {  // (the waves left the previous image at the barrier that ends an item)
      const int4 b = p.blk[it.y];
      const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(p.tri + b.z);
      uint4 *dst = reinterpret_cast<uint4 *>(S.img);
      for (int i = tid; i < (b.w + 3) >> 2; i += NT) dst[i] = src[i];
      for (int i = tid; i < b.y; i += NT) S.rb4[i] = p.rowbase[b.x + i] << 2;
      cur_block = it.y;
      q0 = p.delta + b.x;
      __syncthreads();
    }
    // units it.x + wave, + kCgbWaves, ...: their records sixteen at a time -- lane l holds dword l & 3 of the record of unit k + kCgbWaves * (l >> 2)
    int k = it.x + wave;
    unsigned recv = 0u;
    int rec_k0 = k;  // the unit whose record sits in lanes 0 .. 3
    auto load_recs = [&](const int kf) {
      const long long idx = (long long)kf + (long long)kCgbWaves * (lane >> 2);
      recv = idx < uend ? reinterpret_cast<const unsigned *>(p.units)[idx * 4 + (lane & 3)] : 0u;
      rec_k0 = kf;
    };
    auto unit_of = [&](const int kk) {
      const int l = ((kk - rec_k0) / kCgbWaves) << 2;
      CgbUnit u;
      u.px = (unsigned)readlane((int)recv, l);
      u.mbase = (unsigned)readlane((int)recv, l + 1);
      const unsigned z = (unsigned)readlane((int)recv, l + 2);
      u.d = (int)(z & 0xffffu);
      u.i0 = (int)(z >> 16);
      u.r = readlane((int)recv, l + 3);
      return u;
    };
    // dword `pair * 64 + lane` of the vertex's table (pairs beyond the last one: the last one again -- their tiles are masked)
    auto tab_at = [&](const CgbUnit &u, const int pair) {
      const int npairs = (((u.d + GM_WAVE - 1) >> 6) + 1) >> 1;
      return p.tab[(size_t)u.px + (size_t)(unsigned)((min(pair, npairs - 1) << 6) + lane)];
    };
    // position q of column j (per lane)
    auto q_at = [&](const CgbUnit &u, const int j) {
      const int jj = min(j, u.d - 1);
      const unsigned dw = p.tab[(size_t)u.px + (size_t)(unsigned)(((jj >> 7) << 6) + (jj & (GM_WAVE - 1)))];
      return (jj & GM_WAVE) ? dw >> 16 : dw & 0xffffu;
    };
    auto issue = [&](const CgbUnit &u, CgbHead &h) {
      h.rowq = q_at(u, u.i0 + lane);
      // nine dwords at ONE 32-bit byte offset from the table's (scalar) base + immediates: no clamp -- the table buffer has the slack, and what
      // lies beyond the vertex's last pair belongs to tiles it does not have
      const char *tb = reinterpret_cast<const char *>(p.tab);
      const unsigned off = (u.px + (unsigned)((((u.i0 + 1) >> 7) << 6) + lane)) << 2;
#pragma unroll
      for (int kk = 0; kk < kCgbHead; ++kk) h.hd[kk] = *reinterpret_cast<const unsigned *>(tb + off + (unsigned)(kk << 8));
    };
    auto finish = [&](const CgbUnit &u, const CgbHead &h) {
      const int stride = (u.d + 31) >> 5, ntiles = (u.d + GM_WAVE - 1) >> 6;
      // the vertex's matrix as a buffer resource: a row is `soffset`, a lane's word `voffset` -- and a lane that has no word to store carries
      // kDrop, beyond the resource's range: dropped by the range check, no EXEC mask to set and restore around every store
      const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(p.mat + u.mbase, 0, 0x7fff0000, 0x00020000);
      constexpr int kDrop = (int)0x80000000u;
      int rb4v = S.rb4[min(max((int)h.rowq - q0, 0), kCgbMaxRows - 1)];
      int g = 0;
      while (g < u.r) {
        // a chunk: the rows that share their diagonal tile t0 (and this register of row offsets) -- nearly always the whole unit
        const int i = u.i0 + g, t0 = (i + 1) >> 6;
        if (g && (g & (GM_WAVE - 1)) == 0)  // (units of more than 64 rows: the top blocks)
          rb4v = S.rb4[min(max((int)q_at(u, i + lane) - q0, 0), kCgbMaxRows - 1)];
        const int gend = min(min(u.r, ((t0 + 1) << 6) - 1 - u.i0), (g | (GM_WAVE - 1)) + 1);
        const int soff = i * stride * 4;  // (< 2048 x 64 x 4)
        const int voff_z = lane < 2 * t0 ? lane << 2 : kDrop;
        if (t0 >= ntiles) {  // no column beyond these rows (the last row of a vertex whose d+ is a multiple of 64): zero words, all below 2 t0
          for (int gg = g, so = soff; gg < gend; ++gg, so += stride * 4) __builtin_amdgcn_raw_buffer_store_b32(0u, mrs, voff_z, so, 0);
        }
        int gi = 0;
        for (int tb = t0; tb < ntiles; tb += kCgUnroll, ++gi) {  // the tile groups of the chunk's rows: eight column tiles = sixteen words each
          unsigned pr[5];  // the table dwords of the group's tiles: pairs (tb >> 1) .. + 4
          if (g == 0 && gi == 0) {  // (wave-uniform) out of the registers requested a unit ago
#pragma unroll
            for (int x = 0; x < 5; ++x) pr[x] = h.hd[x];
          } else if (kCgbHead >= 9 && g == 0 && gi == 1) {
#pragma unroll
            for (int x = 0; x < 5; ++x) pr[x] = h.hd[kCgbHead >= 9 ? 4 + x : x];
          } else {  // (rows beyond the head's tiles; a unit whose rows cross a tile)
#pragma unroll
            for (int x = 0; x < 5; ++x) pr[x] = tab_at(u, (tb >> 1) + x);
          }
          const int nt = min(kCgUnroll, ntiles - tb);
          const unsigned long long mlt = (tb + nt == ntiles) ? cgb_mask_lt(u.d, (tb + nt - 1) << 6) : ~0ull;
          const int voff_d = lane < min(2 * kCgUnroll, stride - 2 * tb) ? (2 * tb + lane) << 2 : kDrop;
          const bool head = tb == t0;
          if (tb & 1) cgb_rows_n<1>(nt, S.img, rb4v, pr, g, gend, head, cgb_mask_gt(i, t0 << 6), mlt, mrs, soff, stride * 4, voff_d, voff_z);
          else cgb_rows_n<0>(nt, S.img, rb4v, pr, g, gend, head, cgb_mask_gt(i, t0 << 6), mlt, mrs, soff, stride * 4, voff_d, voff_z);
        }
        g = gend;
      }
    };
#if CGB_DEPTH == 2
    CgbUnit ucur, unxt, unn;
    CgbHead hcur, hnxt, hnn;
    if (k < uend) {
      load_recs(k);
      ucur = unit_of(k);
      issue(ucur, hcur);
      if (k + kCgbWaves < uend) {
        unxt = unit_of(k + kCgbWaves);
        issue(unxt, hnxt);
      }
    }
    while (k < uend) {
      const int k2 = k + 2 * kCgbWaves;
      if (k2 < uend) {
        if (k2 - rec_k0 >= 16 * kCgbWaves) load_recs(k2);
        unn = unit_of(k2);
        issue(unn, hnn);
      }
      finish(ucur, hcur);
      ucur = unxt;
      hcur = hnxt;
      unxt = unn;
      hnxt = hnn;
      k += kCgbWaves;
    }
#else
    CgbUnit ucur, unxt;
    CgbHead hcur, hnxt;
    if (k < uend) {
      load_recs(k);
      ucur = unit_of(k);
      issue(ucur, hcur);
    }
    while (k < uend) {
      const int k1 = k + kCgbWaves;
      if (k1 < uend) {
        if (k1 - rec_k0 >= 16 * kCgbWaves) load_recs(k1);
        unxt = unit_of(k1);
        issue(unxt, hnxt);
      }
      finish(ucur, hcur);
      ucur = unxt;
      hcur = hnxt;
      k = k1;
    }
#endif
    __syncthreads();  // the image and the queue word are rewritten by the next item
  }
}
hipError_t launch_cgatherb(const CGatherBParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(kCgUnroll == 8, "cg_merge moves sixteen lanes; cgb_row dispatches one to eight tiles");
  if (p.tri == nullptr || p.mat == nullptr || p.units == nullptr || p.items == nullptr || p.tab == nullptr) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cgatherb_kernel, dim3((unsigned)grid_blocks), dim3(kCgbWaves * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}
int cgatherb_per_cu() { return (int)std::min<size_t>(163840 / sizeof(CGatherBLds), 2048 / (kCgbWaves * GM_WAVE)); }

hipError_t launch_cgather(const CGatherParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(kCbMaxDeg <= 2048, "a row is at most 64 words: one per lane");
  if (p.core == nullptr || p.mat == nullptr || p.core_bytes == 0 || p.core_bytes > 0xffffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cgather_kernel, dim3((unsigned)grid_blocks), dim3(kCgWaves * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}
// (Round 5 tried the core rows SHARED OUT to the XCDs once more -- the workgroups of XCD k gather only the rows whose core row is
// = k mod 8, a work unit = a batch of 2 .. 8 matrices x one share with their column tables together in LDS and the share's rows listed
// once: traffic 53.6 -> 36.0 GB, L2 hit rate 0.27 -> 0.44 as in round 4, but 5.9 instead of 2.9 G vector instructions and 13.3 instead
// of 11.9 ms -- with a third less traffic the kernel is bound by its instruction stream and the latency of a row's gathers, not by
// lines.  profiles/r05/ab_clique4_gather_shares.txt; not in the tree.)
// workgroups per CU the launch asks for (GM_CG_PER_CU: sweeps).  The gathers are bound by the lines they pull through L2, yet they want
// every wave a CU has: 4 / 2 workgroups per CU 27.7 / 35.6 ms for the whole pattern.
int cgather_per_cu() {
  static const int v = [] {
    const char *e = gm_sweep_env("GM_CG_PER_CU");
    const int cap = (int)std::min<size_t>(163840 / sizeof(CGatherLds), 2048 / (kCgWaves * GM_WAVE));
    return e ? std::max(1, std::min(atoi(e), cap)) : cap;
  }();
  return v;
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_cgather_kernel() {}
void gm_touch_cgather() { hipLaunchKernelGGL(gm_touch_cgather_kernel, dim3(1), dim3(1), 0, 0); }
