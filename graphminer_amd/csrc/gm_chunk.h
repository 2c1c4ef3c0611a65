// gm_chunk.h -- the mining kernel as a template: per-workgroup LDS state, the k-clique level-2 / deeper-level counters on the
// bit-matrix, process_chunk (one task chunk: stage, filter, batches, flattened passes, pattern hooks) and mine_kernel
// (persistent workgroups dequeuing chunks). Instantiated per (pattern, workgroup class) in gm_mine.hip (class 0) and
// gm_mine_wide.hip (the big-LDS classes of the symmetric-graph patterns). See the header comment of gm_mine.hip.
#pragma once
#include "gm_flat.h"

#ifndef GM_EDESC
#define GM_EDESC 1  // 0: A/B build that gathers rp[v], rp[v+1] per task edge instead of reading the edge descriptors
#endif

namespace gm {

#define GM_IS_CLIQUE(P) ((P) == PAT_CLIQUE4 || (P) == PAT_CLIQUEK || (P) == PAT_CLIQUEK_DEEP)
#define GM_IS_PEREDGE(P) ((P) == PAT_DIAMOND || (P) == PAT_MOTIF4E || (P) == PAT_DAGSTATS)  // need |N(v0) ^ N(v1)| per task edge

// Workgroup classes of the mining kernel. CLS 0 is the general one (4 waves, 21-32 KB of LDS, 5-7 workgroups per CU).
// The symmetric-graph patterns add two BIG-LDS classes for the rows just above the 3072-entry stage, which round 1 cut into
// SPLIT chunks whose streamed keys were verified against a dense per-row bitmap in HBM (2 MB per row at nv = 2^24: cold for
// every probe, 4.5 ns per key against 1.2 ns for a staged row -- 88 % of the 3-motif time on R-MAT-24):
//   CLS 1 "mid": rows of 3073..8191 entries, staged whole (32 KB) behind a 2^16-bit filter, 4 waves, 3 workgroups per CU;
//   CLS 2 "big": rows of 8192..24576 entries, staged whole (96 KB) behind a 2^17-bit filter, 16 waves, one workgroup per CU.
// Their chunks are whole single rows (cut into PARTS by estimated work), so they go through the ordinary staged path of
// process_chunk: LDS filter -> candidate queue -> bisection in LDS, no bitmap, no HBM probe.
// class 1 keeps two workgroups per CU: 8 waves each (63 KB) = 16 waves per CU for the 3-motif kernels (91 VGPRs); 12 waves
// each (74 KB) = 24 per CU for diamond, whose 80 VGPRs allow six waves per SIMD.  (4 waves x three workgroups = 12 per CU was
// 11 % slower on R-MAT-24; 10 waves -- not a multiple of the four SIMDs -- 12 % slower than 8: profiles/r02/ab_mid_waves.log.)
constexpr __host__ __device__ int mid_waves_of(int pat) { return pat == PAT_DIAMOND ? 12 : 8; }
template <int PAT, int CLS>
struct MineCfg {
  static constexpr int waves = CLS == 2 ? 16 : (CLS == 1 ? mid_waves_of(PAT) : kWavesPerBlock);
  static constexpr int stage = CLS == 0 ? stage_cap_of(PAT) : CLS == 1 ? kStageCapMid : kStageCapBig;
  static constexpr int fl2 = CLS == 0 ? kFilterLog2 : CLS == 1 ? 16 : 17;
  static constexpr bool multi_row = CLS == 0;  // the wide classes stage ONE row: no per-entry local row table
};

// per-workgroup state of the current task chunk, shared by the waves of the workgroup
template <int PAT, int CLS = 0>
struct alignas(16) BlockLds {
  using Cfg = MineCfg<PAT, CLS>;
  int stage[Cfg::stage];        // staged adjacency slice col[e_begin .. e_end) (first member: the branch-free bisection may
                                // read past the list, never past the workgroup's LDS)
  int rpl[kMaxChunkVerts + 8];  // row offsets of the chunk's vertices (absolute)
  unsigned bits[GM_IS_CLIQUE(PAT) ? kBitWords : 4];
  unsigned fbits[(1 << Cfg::fl2) / 32];   // hashed membership filter over (row, neighbour) pairs of the staged slice
  unsigned char lrow[Cfg::multi_row ? Cfg::stage : 4];  // local row of every staged entry
  int next_batch;               // dynamic batch counter of the chunk
  unsigned queue_pos;           // broadcast slot of the chunk dequeue
  WaveLds w[Cfg::waves];
};

// sum_i sum_{j in M[i]} popc(M[i] & M[j])  ==  sum_{(v0,v1)} sum_{v2 in S1} |S1 ^ N+(v2)|
// (the second DFS level of clique4_warp_edge.cuh:22-27 on the LDS / scratch bit-matrix)
__device__ __forceinline__ unsigned long long clique4_count(const int *__restrict__ rpl, const unsigned *__restrict__ bits,
                                                            const int tid, const int nthreads, const int eb, const int nel,
                                                            const int nvl, const int stride) {
  unsigned long long c = 0;
  for (int le = tid; le < nel; le += nthreads) {
    const int e = eb + le;
    int lo = 0, hi = nvl - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int row0 = rpl[lo] - eb;  // local index of edge (u, A[0])
    const unsigned *Mi = bits + (size_t)le * stride;
    for (int w = 0; w < stride; ++w) {
      unsigned x = Mi[w];
      while (x) {
        const int bit = __ffs((int)x) - 1;
        x &= x - 1;
        const unsigned *Mj = bits + (size_t)(row0 + w * 32 + bit) * stride;
        for (int w2 = 0; w2 < stride; ++w2) c += (unsigned)__popc(Mi[w2] & Mj[w2]);
      }
    }
  }
  return c;
}

// Same sum for a matrix that lives in the global scratch arena (one big vertex, rows 0..nel-1, row0 = 0):
// one wave per row i, lane w holds word w of M_i, the set bits j of M_i are walked with scalar code and the
// rows M_j are fetched with coalesced loads, four independent loads in flight.
__device__ __forceinline__ unsigned long long clique4_count_wide(WaveLds &L, const unsigned *__restrict__ bits, const int lane,
                                                                 const int wave, const int nel, const int stride) {
  // requires stride <= 64 (rows up to 2048 columns).
  // Per row i: the set-bit positions of M_i are expanded into a per-wave LDS list (popcount + DPP scan give each lane its
  // slot), then the rows M_j are fetched G at a time by 64/G-lane groups with 16 independent loads in flight per lane --
  // the matrix of a big vertex lives in the scratch arena (L2 / Infinity Cache), so memory-level parallelism is what counts.
  unsigned short *plist = reinterpret_cast<unsigned short *>(&L);  // the flat-pass scratch is idle during this phase
  constexpr int kCap = (int)(sizeof(WaveLds) / sizeof(unsigned short));
  const int P2 = stride <= 8 ? 8 : stride <= 16 ? 16 : stride <= 32 ? 32 : 64;
  const int G = 64 / P2, gid = lane / P2, wq = lane % P2;
  const bool actw = wq < stride;
  unsigned long long c = 0;
  for (int i = wave; i < nel; i += kWavesPerBlock) {
    const unsigned mi = (lane < stride) ? bits[(size_t)i * stride + lane] : 0u;
    const int cw = __popc(mi);
    const int incl = wave_incl_scan_add(cw);
    const int total = readlane(incl, GM_WAVE - 1);
    if (total == 0) continue;
    if (total <= kCap) {
      unsigned x = mi;
      int k = incl - cw;
      while (x) {
        plist[k++] = (unsigned short)(lane * 32 + (__ffs((int)x) - 1));
        x &= x - 1;
      }
      wave_sync();
      const unsigned mrep = actw ? bits[(size_t)i * stride + wq] : 0u;
      constexpr int kInFlight = 16;
      for (int p = 0; p < total; p += kInFlight * G) {
        unsigned m[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
          const int idx = p + u * G + gid;
          const int j = (idx < total) ? (int)plist[idx] : -1;
          m[u] = (j >= 0 && actw) ? bits[(size_t)j * stride + wq] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) c += (unsigned)__popc(mrep & m[u]);
      }
      wave_sync();
    } else {  // more set bits than the list holds: walk them with scalar code
      for (int w = 0; w < stride; ++w) {
        unsigned x = (unsigned)readlane((int)mi, w);
        while (x) {
          const int bit = __ffs((int)x) - 1;
          x &= x - 1;
          const unsigned mj = (lane < stride) ? bits[(size_t)(w * 32 + bit) * stride + lane] : 0u;
          c += (unsigned)__popc(mi & mj);
        }
      }
    }
  }
  return c;
}

// The same sum with the matrix rows M_j served from LDS: the arena matrix is walked in TILES of TR rows (TR * stride <= 2048
// words, TR a multiple of 32 so that a tile's columns are whole words of M_i); a tile is copied to LDS once (contiguous,
// coalesced), then every row i is read once per tile (4 rows in flight per wave) and only its bits inside the tile's column
// range are walked -- each hit is one ds_read_b32 per lane instead of a global row fetch. Row fetches from the arena drop from
// sum_i |M_i| to nel * nel / TR.
__device__ __forceinline__ unsigned long long clique4_count_tiled(unsigned *__restrict__ tile, const unsigned *__restrict__ gbits,
                                                                  const int tid, const int lane, const int wave, const int nel,
                                                                  const int stride) {
  constexpr int R = 4;
  const int TR = (kBitWords / stride) & ~31;  // stride <= 64  =>  TR >= 32
  const int lw = min(lane, stride - 1);
  const bool actl = lane < stride;
  unsigned long long c = 0;
  for (int t0 = 0; t0 < nel; t0 += TR) {
    const int tr = min(TR, nel - t0);
    __syncthreads();  // the previous tile is no longer read
    for (int i = tid; i < tr * stride; i += kWavesPerBlock * GM_WAVE) tile[i] = gbits[(size_t)t0 * stride + i];
    __syncthreads();
    const int w0 = t0 >> 5, w1 = (t0 + tr + 31) >> 5;  // words of M_i that hold the tile's columns
    for (int ib = wave * R; ib < nel; ib += kWavesPerBlock * R) {
      unsigned mr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = ib + r;
        const unsigned v = gbits[(size_t)min(i, nel - 1) * stride + lw];  // unconditional load, masked below
        mr[r] = (actl && i < nel) ? v : 0u;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        unsigned part = 0;
        for (int ww = w0; ww < w1; ++ww) {
          unsigned x = (unsigned)readlane((int)mr[r], ww);  // wave-uniform: the bits of M_i in columns [32 ww, 32 ww + 32)
          while (x) {  // (four bits per trip with four reads in flight measured slower: 234.8 vs 221.2 ms)
            const int bit = __ffs((int)x) - 1;
            x &= x - 1;
            const unsigned mj = tile[(ww * 32 + bit - t0) * stride + lw];
            part += (unsigned)__popc(mr[r] & mj);
          }
        }
        c += part;
      }
    }
  }
  __syncthreads();
  return c;
}

// ---- k-clique, k >= 5: deeper DFS levels on the same bit-matrix ----------------------------------------
// C_1(S) = |S|,  C_m(S) = sum_{j in S} C_{m-1}(S & M_j);  k-cliques through edge i = C_{k-2}(M_i)
// (the nested intersect levels of clique5..8_warp_edge.cuh / automine_5clique, automine_omp.h:138-157).
constexpr int kSmallWords = 8;  // LDS-resident matrices have rows of <= 256 columns

// Levels up to kCliqueInlineM (k <= 8, the range of the reference's GPU solver: src/clique/gpu_base.cu:59-71) are inlined into one
// another; a deeper level (k = 9..12, counted by the reference's generic clique_omp_recursive / edge_warp_iterative.cuh:2-75) is a real
// function, compiled once and called by every k above it -- inlined all the way the four extra cases took the build from 4 to 16
// minutes (every level carries a copy of all the levels below it, for every k).  The called levels see generic pointers (flat loads).
constexpr int kCliqueInlineM = 6;
struct SmallSet { unsigned w[kSmallWords]; };
template <int M>
__device__ __noinline__ unsigned long long clique_small_outlined(SmallSet S, const unsigned *bits, int row0, int stride);
template <int M>
__device__ __forceinline__ unsigned long long clique_small_call(const unsigned (&S)[kSmallWords], const unsigned *__restrict__ bits,
                                                                const int row0, const int stride);

template <int M>
struct CliqueSmall {  // one lane per row, the candidate set lives in 8 registers
  static __device__ __forceinline__ unsigned long long run(const unsigned (&S)[kSmallWords], const unsigned *__restrict__ bits,
                                                           const int row0, const int stride) {
    unsigned long long c = 0;
#pragma unroll
    for (int w = 0; w < kSmallWords; ++w) {
      unsigned x = S[w];
      while (x) {
        const int bit = __ffs((int)x) - 1;
        x &= x - 1;
        const unsigned *Mj = bits + (size_t)(row0 + w * 32 + bit) * stride;
        unsigned T[kSmallWords];
#pragma unroll
        for (int w2 = 0; w2 < kSmallWords; ++w2) T[w2] = (w2 < stride) ? (S[w2] & Mj[w2]) : 0u;
        c += clique_small_call<M - 1>(T, bits, row0, stride);
      }
    }
    return c;
  }
};
template <>
struct CliqueSmall<1> {
  static __device__ __forceinline__ unsigned long long run(const unsigned (&S)[kSmallWords], const unsigned *, int, int) {
    unsigned c = 0;
#pragma unroll
    for (int w = 0; w < kSmallWords; ++w) c += (unsigned)__popc(S[w]);
    return c;
  }
};

template <int M>
__device__ __noinline__ unsigned long long clique_small_outlined(SmallSet S, const unsigned *bits, int row0, int stride) {
  return CliqueSmall<M>::run(S.w, bits, row0, stride);
}
template <int M>
__device__ __forceinline__ unsigned long long clique_small_call(const unsigned (&S)[kSmallWords], const unsigned *__restrict__ bits,
                                                                const int row0, const int stride) {
  if constexpr (M > kCliqueInlineM) {
    SmallSet s;
#pragma unroll
    for (int w = 0; w < kSmallWords; ++w) s.w[w] = S[w];
    return clique_small_outlined<M>(s, bits, row0, stride);
  } else {
    return CliqueSmall<M>::run(S, bits, row0, stride);
  }
}

template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_small(const int *__restrict__ rpl, const unsigned *__restrict__ bits,
                                                                  const int tid, const int nthreads, const int eb, const int nel,
                                                                  const int nvl, const int stride) {
  unsigned long long c = 0;
  for (int le = tid; le < nel; le += nthreads) {
    const int e = eb + le;
    int lo = 0, hi = nvl - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (rpl[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int row0 = rpl[lo] - eb;
    unsigned S[kSmallWords];
#pragma unroll
    for (int w = 0; w < kSmallWords; ++w) S[w] = (w < stride) ? bits[(size_t)le * stride + w] : 0u;
    c += clique_small_call<M>(S, bits, row0, stride);
  }
  return c;
}

template <int M>
__device__ __noinline__ unsigned long long clique_wide_outlined(unsigned S, const unsigned *bits, int lane, int stride);
template <int M>
__device__ __forceinline__ unsigned long long clique_wide_call(const unsigned S, const unsigned *__restrict__ bits, const int lane, const int stride);

template <int M>
struct CliqueWide {  // one wave per row, lane w holds word w of the candidate set (stride <= 64)
  static __device__ __forceinline__ unsigned long long run(const unsigned S, const unsigned *__restrict__ bits, const int lane,
                                                           const int stride) {
    // the rows M_j of up to four set bits are requested together (unconditional loads; lanes >= stride hold 0 in S, so
    // whatever they read is masked by the AND): one dependent arena round trip per four sub-trees instead of per sub-tree --
    // chunk timings: 94 % of the 5-clique time was this walk, one load at a time
    unsigned long long c = 0;
    const int lw = min(lane, stride - 1);
    for (int w = 0; w < stride; ++w) {
      unsigned x = (unsigned)readlane((int)S, w);  // wave-uniform
      while (x) {
        int j[4];
        int n = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          j[u] = w * 32;
          if (x) {
            j[u] += __ffs((int)x) - 1;
            x &= x - 1;
            n = u + 1;
          }
        }
        unsigned mj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) mj[u] = bits[(size_t)j[u] * stride + lw];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < n) c += clique_wide_call<M - 1>(S & mj[u], bits, lane, stride);
      }
    }
    return c;
  }
};
template <>
struct CliqueWide<1> {
  static __device__ __forceinline__ unsigned long long run(const unsigned S, const unsigned *, int, int) {
    return (unsigned long long)__popc(S);  // per-lane partial; summed over the wave at kernel end
  }
};

template <int M>
__device__ __noinline__ unsigned long long clique_wide_outlined(unsigned S, const unsigned *bits, int lane, int stride) {
  return CliqueWide<M>::run(S, bits, lane, stride);
}
template <int M>
__device__ __forceinline__ unsigned long long clique_wide_call(const unsigned S, const unsigned *__restrict__ bits, const int lane, const int stride) {
  if constexpr (M > kCliqueInlineM) return clique_wide_outlined<M>(S, bits, lane, stride);
  else return CliqueWide<M>::run(S, bits, lane, stride);
}

// k >= 5 on a big vertex, through INDUCED SUB-MATRICES: every deeper level of row i only ever looks at the rows and
// columns in M_i, so the workgroup compacts that |M_i| x |M_i| sub-matrix into LDS once (row j = M_j restricted to the
// columns of M_i, re-indexed by position: a gather of the row's bits + one ballot per 64 columns) and the remaining
// k - 3 levels run on it like on any LDS-resident matrix (one thread per row, the candidate set in 8 registers). The arena is
// read |M_i| rows per row i instead of once per visited sub-tree (chunk timings: that walk was 94 % of the 5-clique time,
// bound by 17 G row fetches of 128 B). Rows with more than 256 set bits: the compacted matrix goes to the next arena slot and is
// processed one level down the same way (k = 5: what is left is a pair count, the 4-clique tile walk).
// pair count of a matrix in the arena whose rows are wider than the 64-lane sweep of the tile walk (> 2048 columns): one wave
// per row i, lanes stride the words; the rows M_j of its set bits are fetched from the arena. Only the top rows of a DAG ever
// get here (a vertex with more than 2048 common out-neighbours with one of its out-neighbours).
__device__ __forceinline__ unsigned long long clique4_count_anywidth(const unsigned *__restrict__ gbits, const int lane, const int wave,
                                                                     const int nel, const int stride) {
  unsigned long long c = 0;
  for (int i = wave; i < nel; i += kWavesPerBlock) {
    const unsigned *Mi = gbits + (size_t)i * stride;
    for (int w = 0; w < stride; ++w) {
      unsigned x = Mi[w];  // (wave-uniform address: a broadcast load)
      while (x) {
        const int j = w * 32 + (__ffs((int)x) - 1);
        x &= x - 1;
        const unsigned *Mj = gbits + (size_t)j * stride;
        unsigned part = 0;
        for (int w2 = lane; w2 < stride; w2 += GM_WAVE) part += (unsigned)__popc(Mi[w2] & Mj[w2]);
        c += part;
      }
    }
  }
  return c;
}

constexpr int kSubMaxStride = 128;  // cliquek_count_sub: rows of up to 4096 columns, two words per lane
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub(unsigned *__restrict__ sub, int *__restrict__ lds_scratch,
                                                                unsigned short *__restrict__ plist, const unsigned *__restrict__ gbits,
                                                                unsigned *__restrict__ sub_arena, const size_t arena_step,
                                                                const int tid, const int lane, const int wave, const int nel,
                                                                const int stride, const bool force_anywidth = false);
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub_body(unsigned *__restrict__ sub, int *__restrict__ lds_scratch,
                                                                unsigned short *__restrict__ plist, const unsigned *__restrict__ gbits,
                                                                unsigned *__restrict__ sub_arena, const size_t arena_step,
                                                                const int tid, const int lane, const int wave, const int nel,
                                                                const int stride, const bool force_anywidth) {
  static_assert(kBitWords >= 256 * kSmallWords, "the sub-matrix of 256 rows must fit the bit-matrix LDS");
  static_assert(sizeof(WaveLds) * kWavesPerBlock >= 2 * 32 * kSubMaxStride, "the position list (one entry per column) lives in the idle pass scratch");
  static_assert(kStageCapClique >= kWavesPerBlock * kSubMaxStride, "one row buffer per wave in the idle stage");
  constexpr int W = kSubMaxStride / GM_WAVE;  // words of a row per lane
  unsigned *rowbuf = reinterpret_cast<unsigned *>(lds_scratch) + wave * kSubMaxStride;  // one row per wave (lds_scratch: 4 x 128 words)
  unsigned long long c = 0;
  for (int i = 0; i < nel; ++i) {
    unsigned mi[W];
    int cw[W], off[W];
    int m = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {  // (all four waves read the row: m is uniform)
      mi[w] = (w * GM_WAVE + lane < stride) ? gbits[(size_t)i * stride + w * GM_WAVE + lane] : 0u;
      cw[w] = __popc(mi[w]);
      const int incl = wave_incl_scan_add(cw[w]);
      off[w] = m + incl - cw[w];  // positions stay ascending: the words 0..63 first, then 64..127
      m += readlane(incl, GM_WAVE - 1);
    }
    if (m == 0) continue;
    const bool in_lds = m <= 256;
    __syncthreads();  // the previous row's sub-matrix / position list is no longer read
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < W; ++w) {
        unsigned x = mi[w];
        int k = off[w];
        while (x) {
          plist[k++] = (unsigned short)((w * GM_WAVE + lane) * 32 + (__ffs((int)x) - 1));
          x &= x - 1;
        }
      }
    }
    __syncthreads();
    // compacted rows: 8 words each in LDS (m <= 256), or `words` words each in the second arena slot (k = 5, wider rows)
    const int words = ((m + 63) >> 6) * 2;
    const int rw = in_lds ? kSmallWords : words;
    unsigned *dst = in_lds ? sub : sub_arena;
    for (int p = wave; p < m; p += kWavesPerBlock) {
      const int j = (int)plist[p];
#pragma unroll
      for (int w = 0; w < W; ++w) rowbuf[w * GM_WAVE + lane] = (w * GM_WAVE + lane < stride) ? gbits[(size_t)j * stride + w * GM_WAVE + lane] : 0u;
      wave_sync();
      for (int q0 = 0; q0 < m; q0 += GM_WAVE) {
        const int q = q0 + lane;
        bool bit = false;
        if (q < m) {
          const int pos = (int)plist[q];
          bit = ((rowbuf[pos >> 5] >> (pos & 31)) & 1u) != 0u;
        }
        const unsigned long long bl = __ballot(bit);
        if (lane == 0) {
          dst[(size_t)p * rw + (q0 >> 5)] = (unsigned)bl;
          dst[(size_t)p * rw + (q0 >> 5) + 1] = (unsigned)(bl >> 32);
        }
      }
      if (in_lds && lane >= words && lane < kSmallWords) sub[p * kSmallWords + lane] = 0u;
      wave_sync();
    }
    if (in_lds) {
      __syncthreads();
      for (int p = tid; p < m; p += kWavesPerBlock * GM_WAVE) {
        unsigned S[kSmallWords];
#pragma unroll
        for (int w = 0; w < kSmallWords; ++w) S[w] = sub[p * kSmallWords + w];
        c += clique_small_call<M - 1>(S, sub, 0, kSmallWords);
      }
    } else {
      __threadfence();
      if constexpr (M == 3) {  // what is left is the pair count of the compacted matrix
        if (words <= GM_WAVE && !force_anywidth) {
          c += clique4_count_tiled(sub, sub_arena, tid, lane, wave, m, words);  // the 4-clique tile walk (begins with a barrier)
        } else {
          __syncthreads();
          c += clique4_count_anywidth(sub_arena, lane, wave, m, words);  // (per-lane partials like the tile walk's)
        }
      } else {  // one level down on the compacted matrix, with the next arena slot for ITS wide rows
        __syncthreads();
        c += cliquek_count_sub<M - 1>(sub, lds_scratch, plist, sub_arena, sub_arena + arena_step, arena_step, tid, lane, wave, m, words, force_anywidth);
      }
    }
  }
  __syncthreads();
  return c;
}

template <int M>
__device__ __noinline__ unsigned long long cliquek_count_sub_outlined(unsigned *sub, int *lds_scratch, unsigned short *plist, const unsigned *gbits,
                                                                      unsigned *sub_arena, size_t arena_step, int tid, int lane, int wave,
                                                                      int nel, int stride, bool force_anywidth) {
  return cliquek_count_sub_body<M>(sub, lds_scratch, plist, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride, force_anywidth);
}
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub(unsigned *__restrict__ sub, int *__restrict__ lds_scratch,
                                                                unsigned short *__restrict__ plist, const unsigned *__restrict__ gbits,
                                                                unsigned *__restrict__ sub_arena, const size_t arena_step,
                                                                const int tid, const int lane, const int wave, const int nel,
                                                                const int stride, const bool force_anywidth) {
  if constexpr (M > kCliqueInlineM)
    return cliquek_count_sub_outlined<M>(sub, lds_scratch, plist, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride, force_anywidth);
  else
    return cliquek_count_sub_body<M>(sub, lds_scratch, plist, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride, force_anywidth);
}

// The same induced-sub-matrix recursion for rows of ANY width (more than kSubMaxStride words: beyond 4096 columns). Nothing of the
// row is held in registers or LDS: the set-bit positions of M_i go to a list in the workgroup's global scratch (plist_g, one list per
// recursion level, plist_step entries apart), the compacted rows are gathered bit by bit from the arena. Slow and exact -- the
// reference's kernels have no width limit (src/clique/gpu_kernels/clique5_warp_edge.cuh:3-39, edge_warp_iterative.cuh:2-75) and
// round 2 refused such rows; only the top rows of a DAG that was not oriented by degree ever get here.
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub_any(unsigned *__restrict__ sub, int *__restrict__ lds_m, int *__restrict__ plist_g,
                                                                    const size_t plist_step, const unsigned *__restrict__ gbits,
                                                                    unsigned *__restrict__ sub_arena, const size_t arena_step, const int tid,
                                                                    const int lane, const int wave, const int nel, const int stride);
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub_any_body(unsigned *__restrict__ sub, int *__restrict__ lds_m, int *__restrict__ plist_g,
                                                                    const size_t plist_step, const unsigned *__restrict__ gbits,
                                                                    unsigned *__restrict__ sub_arena, const size_t arena_step, const int tid,
                                                                    const int lane, const int wave, const int nel, const int stride) {
  unsigned long long c = 0;
  for (int i = 0; i < nel; ++i) {
    __syncthreads();  // the previous row's list / sub-matrix is no longer read
    if (wave == 0) {  // positions of the set bits of M_i, ascending
      int m = 0;
      for (int w0 = 0; w0 < stride; w0 += GM_WAVE) {
        const unsigned word = (w0 + lane < stride) ? gbits[(size_t)i * stride + w0 + lane] : 0u;
        const int cw = __popc(word);
        const int incl = wave_incl_scan_add(cw);
        unsigned x = word;
        int k = m + incl - cw;
        while (x) {
          plist_g[k++] = (w0 + lane) * 32 + (__ffs((int)x) - 1);
          x &= x - 1;
        }
        m += readlane(incl, GM_WAVE - 1);
      }
      if (lane == 0) lds_m[0] = m;
    }
    __threadfence();
    __syncthreads();
    const int m = lds_m[0];
    if (m == 0) continue;
    const bool in_lds = m <= 256;
    const int words = ((m + 63) >> 6) * 2;
    const int rw = in_lds ? kSmallWords : words;
    unsigned *dst = in_lds ? sub : sub_arena;
    for (int p = wave; p < m; p += kWavesPerBlock) {  // row p of the compacted matrix: M_j restricted to the columns of M_i
      const unsigned *Mj = gbits + (size_t)plist_g[p] * stride;
      for (int q0 = 0; q0 < m; q0 += GM_WAVE) {
        const int pos = plist_g[min(q0 + lane, m - 1)];
        const bool bit = (q0 + lane < m) && ((Mj[pos >> 5] >> (pos & 31)) & 1u) != 0u;
        const unsigned long long bl = __ballot(bit);
        if (lane == 0) {
          dst[(size_t)p * rw + (q0 >> 5)] = (unsigned)bl;
          dst[(size_t)p * rw + (q0 >> 5) + 1] = (unsigned)(bl >> 32);
        }
      }
      if (in_lds && lane >= words && lane < kSmallWords) sub[p * kSmallWords + lane] = 0u;
    }
    if (in_lds) {
      __syncthreads();
      for (int p = tid; p < m; p += kWavesPerBlock * GM_WAVE) {
        unsigned S[kSmallWords];
#pragma unroll
        for (int w = 0; w < kSmallWords; ++w) S[w] = sub[p * kSmallWords + w];
        c += clique_small_call<M - 1>(S, sub, 0, kSmallWords);
      }
    } else {
      __threadfence();
      if constexpr (M == 3) {
        if (words <= GM_WAVE) {
          c += clique4_count_tiled(sub, sub_arena, tid, lane, wave, m, words);  // (begins with a barrier)
        } else {
          __syncthreads();
          c += clique4_count_anywidth(sub_arena, lane, wave, m, words);
        }
      } else {
        __syncthreads();
        c += cliquek_count_sub_any<M - 1>(sub, lds_m, plist_g + plist_step, plist_step, sub_arena, sub_arena + arena_step, arena_step, tid, lane, wave, m, words);
      }
    }
  }
  __syncthreads();
  return c;
}

template <int M>
__device__ __noinline__ unsigned long long cliquek_count_sub_any_outlined(unsigned *sub, int *lds_m, int *plist_g, size_t plist_step, const unsigned *gbits,
                                                                          unsigned *sub_arena, size_t arena_step, int tid, int lane, int wave, int nel,
                                                                          int stride) {
  return cliquek_count_sub_any_body<M>(sub, lds_m, plist_g, plist_step, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride);
}
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub_any(unsigned *__restrict__ sub, int *__restrict__ lds_m, int *__restrict__ plist_g,
                                                                    const size_t plist_step, const unsigned *__restrict__ gbits,
                                                                    unsigned *__restrict__ sub_arena, const size_t arena_step, const int tid,
                                                                    const int lane, const int wave, const int nel, const int stride) {
  if constexpr (M > kCliqueInlineM)
    return cliquek_count_sub_any_outlined<M>(sub, lds_m, plist_g, plist_step, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride);
  else
    return cliquek_count_sub_any_body<M>(sub, lds_m, plist_g, plist_step, gbits, sub_arena, arena_step, tid, lane, wave, nel, stride);
}

template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_wide(const unsigned *__restrict__ bits, const int lane, const int wave,
                                                                 const int nel, const int stride) {
  unsigned long long c = 0;
  for (int i = wave; i < nel; i += kWavesPerBlock) {
    const unsigned mi = (lane < stride) ? bits[(size_t)i * stride + lane] : 0u;
    c += clique_wide_call<M>(mi, bits, lane, stride);
  }
  return c;
}

template <int PAT, int CLS>
__device__ __forceinline__ void process_chunk(const MineParams &p, BlockLds<PAT, CLS> &B, const ChunkRec r, const int slot,
                                              const int lane, const int wave, Acc &acc) {
  using Cfg = MineCfg<PAT, CLS>;

  // dense bitmap of the hub row this SPLIT chunk belongs to (nullptr: none was built for it)
  const unsigned *__restrict__ bm = (!GM_IS_CLIQUE(PAT) && slot >= 0 && !(p.flags & 512))
                                        ? p.bitmaps + (size_t)slot * p.bitmap_words : nullptr;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  // (compile-time: with both the descriptor and the rp-gather path alive TC needs 74 VGPRs = 6 waves per SIMD instead of 7;
  // GM_EDESC=0 builds the gather version for A/B runs -- the host passes edesc whenever the graph has entries)
  const int2 *__restrict__ edesc = GM_EDESC ? p.g.edesc : nullptr;
  WaveLds &L = B.w[wave];
  const int tid = threadIdx.x, nthreads = Cfg::waves * GM_WAVE;
  const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
  const int eb = r.e_begin, nel = r.e_end - r.e_begin;

  // ---- workgroup: stage the chunk (coalesced row_ptr / col_idx loads) -----------------------------
  for (int i = tid; i <= nvl; i += nthreads) B.rpl[i] = rp[ub + i];
  if (tid == 0) B.next_batch = 0;
  __syncthreads();
  const bool whole_rows = (eb == B.rpl[0]) && (r.e_end == B.rpl[nvl]);
  if ((p.flags & 128) && !whole_rows) { __syncthreads(); return; }  // ablation: skip SPLIT chunks (counts wrong)
  if ((p.flags & 256) && whole_rows) { __syncthreads(); return; }   // ablation: only SPLIT chunks
  const bool staged = whole_rows && (nel <= Cfg::stage) && !(p.flags & 1);
  const bool use_filter = staged && !(p.flags & 8);
  if (staged) {
    if (use_filter)
      for (int i = tid; i < (1 << Cfg::fl2) / 32; i += nthreads) B.fbits[i] = 0u;
    for (int i = tid; i < nel; i += nthreads) B.stage[i] = col[eb + i];
  }
  // SPLIT chunk of a bitmapped hub row (symmetric-graph patterns): the idle stage becomes a 2^16-bit hashed filter of the
  // whole row, so that ~90 % of the streamed keys are rejected in LDS and only the rest probe the bitmap in HBM
  bool split_filter = false;
  unsigned *sfbits = reinterpret_cast<unsigned *>(B.stage);
  if constexpr (stage_cap_of(PAT) == kStageCapWide && CLS == 0) {
    static_assert(sizeof(B.stage) * 8 >= (1u << kSplitFilterLog2), "stage too small for the SPLIT-row filter");
    split_filter = !whole_rows && bm != nullptr && !(p.flags & 8) && (B.rpl[1] - B.rpl[0]) <= kSplitFilterMaxRow;
    if (split_filter)
      for (int i = tid; i < (1 << kSplitFilterLog2) / 32; i += nthreads) sfbits[i] = 0u;
  }

  // clique: adjacency bit-matrix of the chunk, one row of `stride` words per edge
  int stride = 0;
  bool bits_lds = true, grouped = false;
  int clique_batch = GM_WAVE;  // edges per batch while a big vertex is built in row groups
  int grp_rows = nel > 0 ? nel : 1;
  unsigned *gbits = nullptr;
  if (GM_IS_CLIQUE(PAT)) {
    int m = 0;
    for (int i = lane; i < nvl; i += GM_WAVE) m = max(m, B.rpl[i + 1] - B.rpl[i]);
    stride = (wave_max_nonneg(m) + 31) >> 5;
    const long long words = (long long)nel * stride;
    bits_lds = words <= kBitWords;
    if (bits_lds) {
      for (int i = tid; i < (int)words; i += nthreads) B.bits[i] = 0u;
    } else {
      // a big vertex: its matrix lives in the scratch arena, but it is BUILT in LDS, kGroup rows at a time,
      // and flushed with plain coalesced stores (no device atomics). Only rows wider than the LDS budget
      // (stride > kBitWords/64) fall back to atomics on the arena.
      gbits = p.scratch + (size_t)blockIdx.x * p.scratch_words;
      // (rows of 1025..2048 columns, stride 33..64: 64 rows no longer fit the 2048-word budget -- their groups are 32 rows
      // and their batches 32 edges, which keeps them off the device-atomic path)
      clique_batch = (stride > 32) ? 32 : GM_WAVE;
      const int r = (kBitWords / stride) & ~(clique_batch - 1);
      if (r >= clique_batch) {
        grouped = true;
        grp_rows = r;
      } else {
        clique_batch = GM_WAVE;
        for (long long i = tid; i < words; i += nthreads) gbits[i] = 0u;
      }
    }
  }
  __syncthreads();
  if (split_filter) {
    for (int i = B.rpl[0] + tid; i < B.rpl[1]; i += nthreads) {
      const unsigned h = filter_hash<kSplitFilterLog2>(col[i], 0u);
      atomicOr(&sfbits[h >> 5], 1u << (h & 31u));
    }
    __syncthreads();
  }
  if (staged) {  // local row of every staged entry (+ its filter bit)
    for (int i = tid; i < nel; i += nthreads) {
      const int e = eb + i;
      int lo = 0, hi = nvl - 1;  // owner row: largest r with rpl[r] <= e
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (B.rpl[mid] <= e) lo = mid; else hi = mid - 1;
      }
      if constexpr (Cfg::multi_row) B.lrow[i] = (unsigned char)lo;
      if (use_filter) {
        const unsigned h = filter_hash<Cfg::fl2>(B.stage[i], filter_salt(lo));
        atomicOr(&B.fbits[h >> 5], 1u << (h & 31u));
      }
    }
    __syncthreads();
  }

  // ---- waves: take batches of 64 edges dynamically ------------------------------------------------
  for (int g0 = 0; g0 < nel; g0 += grp_rows) {  // one trip unless a big clique vertex is built in row groups
  const int gend = min(nel, g0 + grp_rows);
  if (GM_IS_CLIQUE(PAT) && grouped) {
    for (int i = tid; i < (gend - g0) * stride; i += nthreads) B.bits[i] = 0u;
    if (tid == 0) B.next_batch = g0 / clique_batch;
    __syncthreads();
  }
  // A big clique vertex is built in row groups that may hold a single batch: then all 4 waves work on EVERY batch of
  // the group, each owning the edges with (lane & 3) == wave, instead of one wave working while three idle.
  const bool split4 = GM_IS_CLIQUE(PAT) && grouped;
  const int bsz = split4 ? clique_batch : r.batch;  // edges per batch (host: 64, or kSplitBatch in heavy SPLIT chunks)
  int my_bi = g0 / bsz;
  auto grab_batch = [&]() {
    int bi = 0;
    if (split4) {
      bi = my_bi++;
    } else {
      if (lane == 0) bi = atomicAdd(&B.next_batch, 1);
      bi = readfirst(bi) * r.nparts + r.part;  // this part's batches (nparts == 1: all of them)
    }
    return bi;
  };
  // GM_DESC_PREFETCH=1 requests the edge descriptors of the NEXT batch before the current batch is processed (the wave takes
  // its next batch index early). Measured on MI355X (profiles/r02/ab_edesc.log), TC ms, descriptors without / with the
  // prefetch: uniform LiveJournal-size 0.885 / 0.887, power-law 1.977 / 1.978, R-MAT-22 10.27 / 10.34 -- the coalesced
  // descriptor load is not what a batch waits for, so the default is 0 (3 VGPRs fewer).
  auto load_desc = [&](const int bi) {
    const int e = eb + bi * bsz + lane;  // (ne < 2^31)
    return edesc[min(e, p.g.ne - 1)];    // unconditional load, clamped index (invalid lanes are masked later)
  };
#ifndef GM_DESC_PREFETCH
#define GM_DESC_PREFETCH 0
#endif
  // (the prefetch holds 3 more VGPRs across the passes; GM_DESC_PREFETCH=0 loads the descriptors at the top of their own batch)
  constexpr bool kPrefetch = GM_DESC_PREFETCH != 0 && GM_EDESC != 0;
  int next_bi = grab_batch();
  int2 next_desc = make_int2(0, 0);
  if (kPrefetch && GM_EDESC && next_bi * bsz < gend) next_desc = load_desc(next_bi);
  for (;;) {
    const int bi = next_bi;
    int2 desc = next_desc;
    const int le0 = bi * bsz;
    if (le0 >= gend) break;
    if (kPrefetch) {
      next_bi = grab_batch();
      if (GM_EDESC && next_bi * bsz < gend) next_desc = load_desc(next_bi);  // wave-uniform condition
    } else if (GM_EDESC) {
      desc = load_desc(bi);
    }
    const int le = le0 + lane;
    const bool valid = (le < nel) && (lane < bsz) && (!split4 || (lane & 3) == wave);
    const int e = eb + le;
    int v = 0, u = 0, ru = 0, a = 0, rv = 0, b = 0, idx = 0, lrow_of_lane = 0;
    if (valid) {
      int lo = 0;
      if (staged) {
        v = B.stage[le];
        lo = Cfg::multi_row ? (int)B.lrow[le] : 0;
      } else {
        v = col[e];
        int hi = nvl - 1;  // owner row: largest i with rpl[i] <= e
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (B.rpl[mid] <= e) lo = mid; else hi = mid - 1;
        }
      }
      lrow_of_lane = lo;
      ru = B.rpl[lo];
      a = B.rpl[lo + 1] - ru;
      u = ub + lo;
      idx = e - ru;
      if (GM_EDESC) {  // (requested one batch ago, coalesced: lane le read entry eb + le)
        rv = desc.x;
        b = desc.y;
      } else {
        rv = rp[v];
        b = rp[v + 1] - rv;
      }
    }
    // pattern-specific task filter / bounds
    bool act = valid;
    int al = a;       // effective length of A = N(u) (a prefix of the row)
    int flag = 0;
    // The symmetric-graph patterns need every UNDIRECTED edge once (the reference takes v1 < v0, diamond.h:5 /
    // automine_formula.h:27 / automine_base.h:18); which endpoint's row hosts the task does not change any count, so the
    // endpoint with the LONGER row normally does (sym_hosts, gm_mine.h): its row is the staged / bitmapped side and the
    // SHORTER list is the one that is streamed -- and pass Y (bisection in HBM) all but disappears.
    const bool owns = sym_hosts(a, b, u, v, stage_cap_of(PAT));
    const int hi = max(u, v), lo = min(u, v);
    if (PAT == PAT_DIAMOND || PAT == PAT_MOTIF4E) act = valid && owns;
    if (PAT == PAT_MOTIF3) {
      // One bounded intersection per UNDIRECTED edge {lo, hi} serves both directed edges of automine_3motif:
      //   I(hi,lo) = |{w in N(hi)^N(lo) : w < lo}|  and  I(lo,hi) = |{w in N(hi)^N(lo) : w < hi}|.
      // A' = {w in N(u) : w < hi} (bounded(), VertexSet.h:240); every common w < hi counts for I(lo,hi), and for
      // I(hi,lo) and the triangle count when additionally w < lo.  sum idx over ALL directed edges is kept per lane.
      if (valid) acc.c2 += (unsigned long long)idx;  // |{w in N(v0): w < v1}| = position of v1 in its row
      act = valid && owns;
      if (act) {
        if (staged) al = lower_bound(&B.stage[ru - eb], a, hi);
        else if (bm == nullptr) al = lower_bound(col + ru, a, hi);
        // (SPLIT chunk with a bitmap: the row itself is never searched, the bound is applied to the streamed keys -- saves a
        // 13-step bisection of the hub row in HBM per edge)
      }
      L.cnt[lane] = (unsigned)lo;  // read back by the match handler (rare)
    }
    if (PAT == PAT_MOTIF3 && act && b >= kMotifTrimMinList) b = lower_bound(col + rv, b, hi);  // only the keys < hi of N(v) can count: trim B too
    act = act && al > 0 && b > 0;
    // direction: X streams B = N(v) and bisects A; Y takes keys from A and bisects B in HBM
    bool dirx = false;
    int vslot = -1;  // dense bitmap of row v, if it has one: pass Y then probes it instead of bisecting N(v)
    // (only the symmetric-graph instantiations carry this path: DAG rows rarely reach the bitmap threshold, and TC would pay
    // for it with 3 VGPRs = one wave per SIMD)
    constexpr bool kRowBitmaps = stage_cap_of(PAT) == kStageCapWide;
    if (kRowBitmaps && p.row_slot != nullptr && act && !(p.flags & 512)) vslot = p.row_slot[v];
    if (act) {
      if (staged) {
        const float cx = (float)b * (float)(p.cost_x_base + p.cost_x_step * bitlen(al));
        // (a bitmap probe is a random 64 B line from a multi-GB region: measured, pricing it below the bisection makes the
        // rule pick Y far too often -- TC 10.9 -> 38 ms at 4 per key; by default the rule ignores the bitmap)
        const float cy = (float)al * ((vslot >= 0 && p.cost_y_bitmap > 0) ? (float)p.cost_y_bitmap
                                                                           : (float)(p.cost_y_base + p.cost_y_step * bitlen(b)));
        dirx = cx <= cy;
      } else if (bm) {
        dirx = (p.flags & 1024) ? ((float)b <= (float)al * (float)(2 + bitlen(b))) : (b <= al);  // bitmap probe vs lg(b) HBM probes
      } else {
        dirx = b <= al;
      }
    }
    const bool diry = act && !dirx;
    if (GM_IS_PEREDGE(PAT)) {
      L.cnt[lane] = 0u;
      wave_sync();
    }

    auto on_found = [&](bool f, int owner, int kidx, int pos, int fl, int key, bool is_x) {
      if (!f) return;
      if (PAT == PAT_TC) {
        acc.c0 += 1;
      } else if (GM_IS_PEREDGE(PAT)) {
        atomicAdd(&L.cnt[owner], 1u);
        if (PAT == PAT_DAGSTATS) acc.c1 += (unsigned long long)(rp[key + 1] - rp[key]);  // d+(v2) of the common neighbour
      } else if (PAT == PAT_MOTIF3) {
        const unsigned below_v = (key < (int)L.cnt[owner]) ? 1u : 0u;
        acc.c0 += 1u + below_v;    // I(v,u) + I(u,v) contributions of this common neighbour
        acc.c1 += below_v;         // triangle u > v > w, counted once (automine_base.h:18)
      } else if (GM_IS_CLIQUE(PAT)) {
        if (p.flags & 4) { acc.c1 += 1; return; }
        const int cbit = is_x ? pos : kidx;  // position of the common neighbour inside N+(u)
        const int word = (le0 + owner) * stride + (cbit >> 5);
        if (bits_lds) atomicOr(&B.bits[word], 1u << (cbit & 31));
        else if (grouped) atomicOr(&B.bits[word - g0 * stride], 1u << (cbit & 31));
        else atomicOr(&gbits[word], 1u << (cbit & 31));
      }
    };

    // pass X
    {
      int llen = (dirx && !(p.flags & 2048)) ? b : 0;  // (2048: ablation, skip pass X)
      if (PAT == PAT_MOTIF3 && split_filter && llen > 0 && b < 128) llen = lower_bound(col + rv, b, hi);  // (the filtered pass has no key bound)
      const int s_len_flag = al | (flag << 30);
      auto actx = [&](bool f, int owner, int kidx, int pos, int fl, int key) { on_found(f, owner, kidx, pos, fl, key, true); };
      if (use_filter)
        flat_pass_filtered<Cfg::fl2>(L, B.stage, B.fbits, col, lane, llen, rv, (ru - eb) | (int)(filter_salt(lrow_of_lane) << 16),
                           s_len_flag, p.flags, actx);
      else if (staged) flat_pass<SEARCH_LDS>(L, B.stage, col, bm, lane, llen, rv, ru - eb, s_len_flag, actx);
      else if (bm && split_filter)
        flat_pass_filtered<kSplitFilterLog2, true>(L, nullptr, sfbits, col, lane, llen, rv, 0, s_len_flag, p.flags, actx, bm);
      else if (bm) flat_pass<SEARCH_BITMAP>(L, B.stage, col, bm, lane, llen, rv, (PAT == PAT_MOTIF3) ? hi : 0x7fffffff, s_len_flag, actx);
      else flat_pass<SEARCH_HBM>(L, B.stage, col, bm, lane, llen, rv, ru, s_len_flag, actx);
    }
    // pass Y: keys from A bisect B = N(v) in HBM -- or, when v is a hub row with a dense bitmap, probe that (one load
    // instead of ~lg b dependent ones; on symmetric R-MAT graphs most Y keys go against hub rows)
    {
      auto acty = [&](bool f, int owner, int kidx, int pos, int fl, int key) { on_found(f, owner, kidx, pos, fl, key, false); };
      const bool y_on = diry && !(p.flags & 0x10000);  // (0x10000: ablation, skip pass Y)
      flat_pass<SEARCH_HBM>(L, B.stage, col, bm, lane, (y_on && vslot < 0) ? al : 0, ru, rv, b | (flag << 30), acty);
      if (kRowBitmaps && p.row_slot != nullptr)
        flat_pass<SEARCH_BITMAP_ROW>(L, B.stage, col, p.bitmaps, lane, (y_on && vslot >= 0) ? al : 0, ru, 0x7fffffff,
                                     max(vslot, 0) | (flag << 30), acty, p.bitmap_words);
    }

    if (PAT == PAT_DIAMOND) {
      wave_sync();
      const unsigned long long n = L.cnt[lane];
      acc.c0 += n * (n - 1) / 2;  // C(n,2), 64-bit (diamond_count.cuh:15-17)
      wave_sync();
    }
    if (PAT == PAT_DAGSTATS) {
      wave_sync();
      const unsigned long long n = valid ? L.cnt[lane] : 0ull;
      acc.c0 += n * n;
      acc.c2 += n;
      wave_sync();
    }
    if (PAT == PAT_MOTIF4E) {
      // per-edge sums of the formula-based 4-motif (src/motif/cpu_kernels/automine_formula.h:30-39)
      wave_sync();
      if (valid && owns) {
        const unsigned long long tri = L.cnt[lane];
        const unsigned long long su = (unsigned long long)a - tri - 1ull, sv = (unsigned long long)b - tri - 1ull;
        acc.c0 += su * (su - 1ull) + sv * (sv - 1ull);  // counter[0]
        acc.c1 += su * sv;                              // counter[1]
        acc.c2 += tri * (su + sv);                      // counter[2]
        acc.c3 += tri * (tri - 1ull);                   // counter[4]
      }
      wave_sync();
    }
    if (!kPrefetch) next_bi = grab_batch();
  }
  if (GM_IS_CLIQUE(PAT) && grouped) {  // flush the finished rows of this group to the arena
    __syncthreads();
    for (int i = tid; i < (gend - g0) * stride; i += nthreads) gbits[(size_t)g0 * stride + i] = B.bits[i];
    __syncthreads();
  }
  }  // row groups

  __syncthreads();  // every batch of the chunk is done (LDS is reused by the next chunk)
  if (GM_IS_CLIQUE(PAT) && !(p.flags & 2)) {
    const bool wide = !bits_lds && nvl == 1 && stride <= GM_WAVE;        // one 64-lane sweep per row
    const bool wide2 = !bits_lds && nvl == 1 && stride <= kSubMaxStride;  // k >= 5: the sub-matrix path takes two words per lane
    if (((p.flags & 4096) && bits_lds) || ((p.flags & 8192) && !bits_lds)) { __syncthreads(); return; }  // ablation
    if (!bits_lds) {
      __threadfence();  // the scratch matrix was written by all 4 waves (plain stores or device atomics)
      __syncthreads();
    }
    // (the matrix pointer is passed with its address space visible -- B.bits = LDS, gbits = global: through a common
    // generic pointer every load became a FLAT load)
    switch (PAT == PAT_CLIQUE4 ? 4 : p.k) {
      case 4:
        if (wide && !(p.flags & 64)) acc.c0 += clique4_count_tiled(B.bits, gbits, tid, lane, wave, nel, stride);
        else if (wide) acc.c0 += clique4_count_wide(L, gbits, lane, wave, nel, stride);  // (64: the per-pair row fetches, A/B)
        else if (bits_lds) acc.c0 += clique4_count(B.rpl, B.bits, tid, nthreads, eb, nel, nvl, stride);
        else acc.c0 += clique4_count(B.rpl, gbits, tid, nthreads, eb, nel, nvl, stride);
        break;
#define GM_CLIQUE_CASE(K, INSTANCE)                                                                                   \
      case K:                                                                                                \
        if (PAT != INSTANCE) break;                                                                          \
        if (wide2 && !((p.flags & 64) && wide))                                                               \
          acc.c0 += cliquek_count_sub<K - 2>(B.bits, B.stage, reinterpret_cast<unsigned short *>(&B.w[0]), gbits,       \
                                             gbits + p.scratch_region, p.scratch_region, tid, lane,   \
                                             wave, nel, stride, (p.flags & (1 << 20)) != 0);                               \
        else if (wide) acc.c0 += cliquek_count_wide<K - 2>(gbits, lane, wave, nel, stride);                      \
        else if (bits_lds) acc.c0 += cliquek_count_small<K - 2>(B.rpl, B.bits, tid, nthreads, eb, nel, nvl, stride); \
        else if (!bits_lds && nvl == 1) /* a row wider than 4096 columns: everything from the workgroup's global scratch */ \
          acc.c0 += cliquek_count_sub_any<K - 2>(B.bits, &B.next_batch, reinterpret_cast<int *>(gbits + (size_t)(K - 2) * p.scratch_region), \
                                                 (size_t)p.scratch_plist, gbits, gbits + p.scratch_region, p.scratch_region, tid, lane, wave, nel, stride); \
        else acc.c1 += 1; /* (cannot happen: a matrix beyond LDS is a one-row chunk) */                        \
        break;
      GM_CLIQUE_CASE(5, PAT_CLIQUEK)
      GM_CLIQUE_CASE(6, PAT_CLIQUEK)
      GM_CLIQUE_CASE(7, PAT_CLIQUEK)
      GM_CLIQUE_CASE(8, PAT_CLIQUEK)
      // beyond the reference's GPU dispatch (src/clique/gpu_base.cu:59-71 stops at 8; its clique_omp_recursive and
      // edge_warp_iterative.cuh:2-75 are generic in k, src/clique/README.md:59 lists k = 9): an instance of its own, so that the called
      // levels (kCliqueInlineM) and their stack stay out of the kernel that counts k <= 8
      GM_CLIQUE_CASE(9, PAT_CLIQUEK_DEEP)
      GM_CLIQUE_CASE(10, PAT_CLIQUEK_DEEP)
      GM_CLIQUE_CASE(11, PAT_CLIQUEK_DEEP)
      GM_CLIQUE_CASE(12, PAT_CLIQUEK_DEEP)
#undef GM_CLIQUE_CASE
      default: break;
    }
    __syncthreads();
  }
}

#ifndef GM_TC_WAVES
#define GM_TC_WAVES 7
#endif
// (second launch bound = workgroups per CU the register allocator aims for: TC's LDS allows 7, asking for 8 made the allocator
// give up at 74 VGPRs = 6 waves per SIMD once the edge descriptors were added; asking for 7 makes it fit the 72 of 7 waves)
// The symmetric-graph patterns (31.5 KB of LDS: 5 workgroups per CU) get 5 for the same reason: 96 VGPRs, not 97.
template <int PAT, int CLS = 0>
__global__ __launch_bounds__((MineCfg<PAT, CLS>::waves * GM_WAVE), (CLS == 2 ? 1 : CLS == 1 ? 2 : (PAT == PAT_CLIQUEK || PAT == PAT_CLIQUEK_DEEP) ? 4 : (PAT == PAT_TC ? GM_TC_WAVES : 5)))
void mine_kernel(const MineParams p) {
  __shared__ BlockLds<PAT, CLS> B;
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  Acc acc;
  for (;;) {
    if (threadIdx.x == 0) B.queue_pos = atomicAdd(p.queue, (unsigned)p.grab);
    __syncthreads();
    const unsigned q = B.queue_pos;
    if (q >= (unsigned)p.count) break;
    const unsigned qe = min(q + (unsigned)p.grab, (unsigned)p.count);
    for (unsigned i = q; i < qe; ++i) {
      const size_t pos = (size_t)p.first + (size_t)i * (size_t)p.step;
      const size_t cid = p.order ? (size_t)p.order[pos] : pos;
      const ChunkRec r = p.chunks[cid];
      const int slot = p.chunk_slot ? p.chunk_slot[cid] : -1;
#ifdef GM_DEBUG_CHUNKS
      const unsigned long long t0 = wall_clock64();
#endif
      process_chunk<PAT, CLS>(p, B, r, slot, lane, wave, acc);  // ends with a workgroup barrier
#ifdef GM_DEBUG_CHUNKS
      if (p.chunk_ticks && threadIdx.x == 0) p.chunk_ticks[pos] = wall_clock64() - t0;
#endif
    }
  }
  const unsigned long long s0 = wave_sum_u64(acc.c0);
  const unsigned long long s1 = wave_sum_u64(acc.c1);
  const unsigned long long s2 = wave_sum_u64(acc.c2);
  const unsigned long long s3 = wave_sum_u64(acc.c3);
  if (lane == 0) {
    if (s0) atomicAdd(&p.counters[0], s0);
    if (s1) atomicAdd(&p.counters[1], s1);
    if (s2) atomicAdd(&p.counters[2], s2);
    if (s3) atomicAdd(&p.counters[3], s3);
  }
}

}  // namespace gm
