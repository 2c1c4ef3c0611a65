// gm_mine.hip -- the subgraph-matching hot path as hand-written HIP for gfx950 (wave64).
//
// What it replaces (reference, CUDA warp-per-edge kernels; NOT translated):
//   warp_edge                 src/triangle/gpu_kernels/bs_warp_edge.cuh:2-18
//   diamond_warp_edge_count   src/sgl/gpu_kernels/diamond_count.cuh:3-21
//   clique4_warp_edge         src/clique/gpu_kernels/clique4_warp_edge.cuh:3-31
//   motif3_warp_edge          src/motif/gpu_kernels/motif3_edge_warp.cuh:2-23
// and the CPU loops they mirror (src/triangle/omp_base.cc:15-21, src/sgl/cpu_kernels/diamond.h,
// src/clique/cpu_kernels/automine_omp.h:67-83, src/motif/cpu_kernels/automine_base.h:2-22).
//
// MI355X design:
//  * workgroup = 4 waves. Workgroups dequeue TASK CHUNKS (a contiguous vertex range owning up to 1024 CSR
//    entries), stage the chunk's row offsets and adjacency slice in LDS with coalesced loads ("adjacency
//    slices staged in LDS") together with a hashed membership bit filter; the waves then take batches of
//    64 task edges (u,v) from an LDS counter.
//  * the reference gives each edge a whole 32-lane warp, so on LiveJournal (mean oriented list length 8.8)
//    most lanes idle. Here the lookup lists of the 64 edges of a batch are FLATTENED into one virtual
//    array, lane p handles element p; the element->edge map is an owner-mark array in LDS resolved by a
//    DPP max-scan. Long lists (>= 192 keys) are streamed with a wave-uniform descriptor instead.
//  * per edge the cheaper direction is chosen: (X) stream N(v) with coalesced loads and test the keys against
//    the LDS-resident N(u) (bit filter -> ballot compaction -> branch-free bisection of the survivors), or
//    (Y) take the keys from N(u) and bisect N(v) in HBM/L2. Hub rows longer than the stage (SPLIT chunks of
//    symmetric graphs) are probed through a dense vertex-id bitmap.
//  * counts stay in 64-bit per-lane registers; one global atomic per wave per counter at the end.
//  * k-clique keeps the candidate sets as an LDS bit-matrix over N+(u) (row = edge (u,v1), bit j =
//    N+(u)[j] in N+(v1)); deeper DFS levels are popcounts of row ANDs without touching HBM.
#include "gm_flat.h"

#ifndef GM_EDESC
#define GM_EDESC 1  // 0: A/B build that gathers rp[v], rp[v+1] per task edge instead of reading the edge descriptors
#endif

namespace gm {

#define GM_IS_CLIQUE(P) ((P) == PAT_CLIQUE4 || (P) == PAT_CLIQUEK)
#define GM_IS_PEREDGE(P) ((P) == PAT_DIAMOND || (P) == PAT_MOTIF4E || (P) == PAT_DAGSTATS)  // need |N(v0) ^ N(v1)| per task edge

// per-workgroup state of the current task chunk, shared by the 4 waves
template <int PAT>
struct alignas(16) BlockLds {
  int stage[stage_cap_of(PAT)];  // staged adjacency slice col[e_begin .. e_end)
  int rpl[kMaxChunkVerts + 8];  // row offsets of the chunk's vertices (absolute)
  unsigned bits[GM_IS_CLIQUE(PAT) ? kBitWords : 4];
  unsigned fbits[kFilterWords];        // hashed membership filter over (row, neighbour) pairs of the staged slice
  unsigned char lrow[stage_cap_of(PAT)];  // local row of every staged entry
  int next_batch;               // dynamic batch counter of the chunk
  unsigned queue_pos;           // broadcast slot of the chunk dequeue
  WaveLds w[kWavesPerBlock];
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
        c += CliqueSmall<M - 1>::run(T, bits, row0, stride);
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
    c += CliqueSmall<M>::run(S, bits, row0, stride);
  }
  return c;
}

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
          if (u < n) c += CliqueWide<M - 1>::run(S & mj[u], bits, lane, stride);
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

// k >= 5 on a big vertex, through INDUCED SUB-MATRICES: every deeper level of row i only ever looks at the rows and
// columns in M_i, so the workgroup compacts that |M_i| x |M_i| sub-matrix into LDS once (row j = M_j restricted to the
// columns of M_i, re-indexed by position: a gather of the row's bits + one ballot per 64 columns) and the remaining
// k - 3 levels run on it like on any LDS-resident matrix (one thread per row, the candidate set in 8 registers). The arena is
// read |M_i| rows per row i instead of once per visited sub-tree (chunk timings: that walk was 94 % of the 5-clique time,
// bound by 17 G row fetches of 128 B). Rows with more than 256 set bits: the compacted matrix goes to the next arena slot and is
// processed one level down the same way (k = 5: what is left is a pair count, the 4-clique tile walk).
template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_sub(unsigned *__restrict__ sub, int *__restrict__ lds_scratch,
                                                                unsigned short *__restrict__ plist, const unsigned *__restrict__ gbits,
                                                                unsigned *__restrict__ sub_arena, const size_t arena_step,
                                                                const int tid, const int lane, const int wave, const int nel,
                                                                const int stride) {
  static_assert(kBitWords >= 256 * kSmallWords, "the sub-matrix of 256 rows must fit the bit-matrix LDS");
  unsigned *rowbuf = reinterpret_cast<unsigned *>(lds_scratch) + wave * GM_WAVE;  // 64 words per wave
  unsigned long long c = 0;
  for (int i = 0; i < nel; ++i) {
    const unsigned mi = (lane < stride) ? gbits[(size_t)i * stride + lane] : 0u;  // (all four waves read the row: m is uniform)
    const int cw = __popc(mi);
    const int incl = wave_incl_scan_add(cw);
    const int m = readlane(incl, GM_WAVE - 1);
    if (m == 0) continue;
    const bool in_lds = m <= 256;
    __syncthreads();  // the previous row's sub-matrix / position list is no longer read
    if (wave == 0) {
      unsigned x = mi;
      int k = incl - cw;
      while (x) {
        plist[k++] = (unsigned short)(lane * 32 + (__ffs((int)x) - 1));
        x &= x - 1;
      }
    }
    __syncthreads();
    // compacted rows: 8 words each in LDS (m <= 256), or `words` words each in the second arena slot (k = 5, wider rows)
    const int words = ((m + 63) >> 6) * 2;
    const int rw = in_lds ? kSmallWords : words;
    unsigned *dst = in_lds ? sub : sub_arena;
    for (int p = wave; p < m; p += kWavesPerBlock) {
      const int j = (int)plist[p];
      rowbuf[lane] = (lane < stride) ? gbits[(size_t)j * stride + lane] : 0u;
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
        c += CliqueSmall<M - 1>::run(S, sub, 0, kSmallWords);
      }
    } else {
      __threadfence();
      if constexpr (M == 3) {  // what is left is the pair count of the compacted matrix: the 4-clique tile walk (begins with a barrier)
        c += clique4_count_tiled(sub, sub_arena, tid, lane, wave, m, words);
      } else {  // one level down on the compacted matrix, with the next arena slot for ITS wide rows
        __syncthreads();
        c += cliquek_count_sub<M - 1>(sub, lds_scratch, plist, sub_arena, sub_arena + arena_step, arena_step, tid, lane, wave, m, words);
      }
    }
  }
  __syncthreads();
  return c;
}

template <int M>
__device__ __forceinline__ unsigned long long cliquek_count_wide(const unsigned *__restrict__ bits, const int lane, const int wave,
                                                                 const int nel, const int stride) {
  unsigned long long c = 0;
  for (int i = wave; i < nel; i += kWavesPerBlock) {
    const unsigned mi = (lane < stride) ? bits[(size_t)i * stride + lane] : 0u;
    c += CliqueWide<M>::run(mi, bits, lane, stride);
  }
  return c;
}

template <int PAT>
__device__ __forceinline__ void process_chunk(const MineParams &p, BlockLds<PAT> &B, const ChunkRec r, const int slot,
                                              const int lane, const int wave, Acc &acc) {

  // dense bitmap of the hub row this SPLIT chunk belongs to (nullptr: none was built for it)
  const unsigned *__restrict__ bm = (!GM_IS_CLIQUE(PAT) && slot >= 0 && !(p.flags & 512))
                                        ? p.bitmaps + (size_t)slot * p.bitmap_words : nullptr;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  // (compile-time: with both the descriptor and the rp-gather path alive TC needs 74 VGPRs = 6 waves per SIMD instead of 7;
  // GM_EDESC=0 builds the gather version for A/B runs -- the host passes edesc whenever the graph has entries)
  const int2 *__restrict__ edesc = GM_EDESC ? p.g.edesc : nullptr;
  WaveLds &L = B.w[wave];
  const int tid = threadIdx.x, nthreads = kWavesPerBlock * GM_WAVE;
  const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
  const int eb = r.e_begin, nel = r.e_end - r.e_begin;

  // ---- workgroup: stage the chunk (coalesced row_ptr / col_idx loads) -----------------------------
  for (int i = tid; i <= nvl; i += nthreads) B.rpl[i] = rp[ub + i];
  if (tid == 0) B.next_batch = 0;
  __syncthreads();
  const bool whole_rows = (eb == B.rpl[0]) && (r.e_end == B.rpl[nvl]);
  if ((p.flags & 128) && !whole_rows) { __syncthreads(); return; }  // ablation: skip SPLIT chunks (counts wrong)
  if ((p.flags & 256) && whole_rows) { __syncthreads(); return; }   // ablation: only SPLIT chunks
  const bool staged = whole_rows && (nel <= stage_cap_of(PAT)) && !(p.flags & 1);
  const bool use_filter = staged && !(p.flags & 8);
  if (staged) {
    if (use_filter)
      for (int i = tid; i < kFilterWords; i += nthreads) B.fbits[i] = 0u;
    for (int i = tid; i < nel; i += nthreads) B.stage[i] = col[eb + i];
  }
  // SPLIT chunk of a bitmapped hub row (symmetric-graph patterns): the idle stage becomes a 2^16-bit hashed filter of the
  // whole row, so that ~90 % of the streamed keys are rejected in LDS and only the rest probe the bitmap in HBM
  bool split_filter = false;
  unsigned *sfbits = reinterpret_cast<unsigned *>(B.stage);
  if constexpr (stage_cap_of(PAT) == kStageCapWide) {
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
      B.lrow[i] = (unsigned char)lo;
      if (use_filter) {
        const unsigned h = filter_hash(B.stage[i], filter_salt(lo));
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
        lo = (int)B.lrow[le];
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
    if (PAT == PAT_MOTIF3 && act && b >= 128) b = lower_bound(col + rv, b, hi);  // only the keys < hi of N(v) can count: trim B too
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
        flat_pass_filtered(L, B.stage, B.fbits, col, lane, llen, rv, (ru - eb) | (int)(filter_salt(lrow_of_lane) << 16),
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
    const bool wide = !bits_lds && nvl == 1 && stride <= GM_WAVE;
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
#define GM_CLIQUE_CASE(K)                                                                                   \
      case K:                                                                                                \
        if (PAT != PAT_CLIQUEK) break;                                                                       \
        if (wide && !(p.flags & 64))                                                                          \
          acc.c0 += cliquek_count_sub<K - 2>(B.bits, B.stage, reinterpret_cast<unsigned short *>(B.fbits), gbits,       \
                                             gbits + p.scratch_words / (K - 3 + 1), p.scratch_words / (K - 3 + 1), tid, lane,   \
                                             wave, nel, stride);                                                          \
        else if (wide) acc.c0 += cliquek_count_wide<K - 2>(gbits, lane, wave, nel, stride);                      \
        else if (bits_lds) acc.c0 += cliquek_count_small<K - 2>(B.rpl, B.bits, tid, nthreads, eb, nel, nvl, stride); \
        else acc.c1 += 1; /* row wider than 2048 columns: not supported for k >= 5 (reported by the host) */ \
        break;
      GM_CLIQUE_CASE(5)
      GM_CLIQUE_CASE(6)
      GM_CLIQUE_CASE(7)
      GM_CLIQUE_CASE(8)
#undef GM_CLIQUE_CASE
      default: break;
    }
    __syncthreads();
  }
}

template <int PAT>
#ifndef GM_TC_WAVES
#define GM_TC_WAVES 7
#endif
// (second launch bound = workgroups per CU the register allocator aims for: TC's LDS allows 7, asking for 8 made the allocator
// give up at 74 VGPRs = 6 waves per SIMD once the edge descriptors were added; asking for 7 makes it fit the 72 of 7 waves)
// The symmetric-graph patterns (31.5 KB of LDS: 5 workgroups per CU) get 5 for the same reason: 96 VGPRs, not 97.
__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE, PAT == PAT_CLIQUEK ? 4 : (PAT == PAT_TC ? GM_TC_WAVES : 5)) void mine_kernel(const MineParams p) {
  __shared__ BlockLds<PAT> B;
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
      process_chunk<PAT>(p, B, r, slot, lane, wave, acc);  // ends with a workgroup barrier
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

// ---- rectangle (4-cycle), flattened ------------------------------------------------------------------------
// src/sgl/cpu_kernels/rectangle.h:1-11:  for v0, v1 in N(v0) (v1 < v0), v2 in N(v0) (v2 < v1):
//                                            count += |{w in N(v1) ^ N(v2) : w < v0}|
// The task space is the set of WEDGES (v1, v0, v2). A wave takes 64 wedges of one centre v0 (wedge id t ->
// (i, j), j < i < idx0[v0], by inverting the triangular number), trims both neighbour lists to keys < v0 and runs
// ONE flattened pass: the shorter trimmed list is the lookup list, the longer one is bisected in HBM/L2.
__global__ __launch_bounds__(256) void idx0_kernel(GraphView g, int *__restrict__ idx0) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < g.nv) idx0[v] = lower_bound(g.col + g.rp[v], g.rp[v + 1] - g.rp[v], v);
}

// PENT = false: rectangle (one intersection per wedge).  PENT = true: pentagon, src/sgl/cpu_kernels/pentagon.h:2-17 --
// the same wedges (v1, v0, v2); every lane then walks v3 in N(v2) (v3 < v0, v3 != v1) and round r intersects
// N(v1) with the r-th v3 of all 64 wedges in ONE flattened pass, counting common w < v0, w != v2.
template <bool PENT>
__global__ __launch_bounds__(256) void rect_flat_kernel(const RectParams p) {
  __shared__ WaveLds W[kWavesPerBlock];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63;
  WaveLds &L = W[threadIdx.x >> 6];
  unsigned long long cnt = 0;
  for (;;) {
    unsigned long long q = 0;
    if (lane == 0) q = atomicAdd(p.queue, 1ull);
    q = ((unsigned long long)(unsigned)readfirst((int)(q >> 32)) << 32) | (unsigned)readfirst((int)q);
    if (q >= p.count) break;
    const unsigned long long gid = p.first + q * p.step;
    const unsigned long long b0 = gid * (unsigned long long)p.group;
    const unsigned long long b1 = min(p.nblocks, b0 + (unsigned long long)p.group);
    // centre of the first block: largest v with block_prefix[v] <= b0 (wave-uniform bisection)
    int lo = 0, hi = p.g.nv - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (p.block_prefix[mid] <= b0) lo = mid; else hi = mid - 1;
    }
    int v0 = lo;
    for (unsigned long long blk = b0; blk < b1; ++blk) {
      while (p.block_prefix[v0 + 1] <= blk) ++v0;
      const int n0 = p.idx0[v0];
      const unsigned long long nw = (unsigned long long)n0 * (unsigned long long)(n0 - 1) / 2ull;  // wedges of v0
      const unsigned long long t = (blk - p.block_prefix[v0]) * 64ull + (unsigned long long)lane;
      const bool valid = t < nw;
      int v1 = 0, v2 = 0, r1 = 0, d1 = 0, r2 = 0, d2 = 0;
      if (valid) {
        // t = i(i-1)/2 + j, 0 <= j < i  ->  i = floor((1 + sqrt(1 + 8t)) / 2), fixed up against rounding
        long long i = (long long)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
        while (i * (i - 1) / 2 > (long long)t) --i;
        while ((i + 1) * i / 2 <= (long long)t) ++i;
        const long long j = (long long)t - i * (i - 1) / 2;
        const int *A0 = col + rp[v0];
        v1 = A0[i];
        v2 = A0[j];
        r1 = rp[v1];
        r2 = rp[v2];
        d1 = lower_bound(col + r1, rp[v1 + 1] - r1, v0);  // keys >= v0 can never count (rectangle.h:8, pentagon.h:12)
        d2 = lower_bound(col + r2, rp[v2 + 1] - r2, v0);
      }
      if (!PENT) {
        int llen = 0, key_base = 0, s_base = 0, s_len = 0;
        if (valid) {
          if (d1 <= d2) { llen = d1; key_base = r1; s_base = r2; s_len = d2; }
          else { llen = d2; key_base = r2; s_base = r1; s_len = d1; }
          if (s_len == 0) llen = 0;
        }
        auto act = [&](bool f, int, int, int, int, int) { cnt += f ? 1u : 0u; };
        flat_pass<SEARCH_HBM>(L, nullptr, col, nullptr, lane, llen, key_base, s_base, s_len, act);
      } else {
        L.cnt[lane] = (unsigned)v2;  // the excluded ancestor of this wedge (intersection_num_bound_except(..., v0, v2))
        wave_sync();
        const int rounds = wave_max_nonneg(valid ? d2 : 0);  // v3 candidates: the d2 entries of N(v2) below v0
        for (int r = 0; r < rounds; ++r) {
          int llen = 0, key_base = 0, s_base = 0, s_len = 0;
          if (valid && r < d2 && d1 > 0) {
            const int v3 = col[r2 + r];
            if (v3 != v1) {
              const int r3 = rp[v3];
              const int d3 = lower_bound(col + r3, rp[v3 + 1] - r3, v0);
              if (d1 <= d3) { llen = d1; key_base = r1; s_base = r3; s_len = d3; }
              else { llen = d3; key_base = r3; s_base = r1; s_len = d1; }
              if (s_len == 0) llen = 0;
            }
          }
          auto act = [&](bool f, int owner, int, int, int, int key) { cnt += (f && key != (int)L.cnt[owner]) ? 1u : 0u; };
          flat_pass<SEARCH_HBM>(L, nullptr, col, nullptr, lane, llen, key_base, s_base, s_len, act);
        }
        wave_sync();
      }
    }
  }
  const unsigned long long s0 = wave_sum_u64(cnt);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

// ---- rectangle by wedge accumulation ---------------------------------------------------------------------------
// rectangle.h:1-11 counts, for the largest vertex v0 of a 4-cycle, sum_{v2 < v1 in N(v0), both < v0} |{w in N(v1) ^ N(v2) : w < v0}|
//   = sum_{w < v0} C(c(w), 2),   c(w) = |{x in N(v0) ^ N(w) : x < v0}|
// so one walk over the 2-paths v0 - x - w (x, w < v0) with a vertex-indexed counter map is enough: every increment adds the
// counter's OLD value (0 + 1 + ... + (c-1) = C(c,2)); a second walk over the same 2-paths clears the map. No intersections.
// A wave owns a map; light centres go one per wave, heavy centres (many 2-paths) use all 4 waves on wave 0's map.
__global__ __launch_bounds__(256) void rect_work_kernel(GraphView g, const int *__restrict__ idx0, unsigned long long *__restrict__ work) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.nv) return;
  unsigned long long w = 0;
  const int r0 = g.rp[v];
  for (int i = 0; i < idx0[v]; ++i) {
    const int x = g.col[r0 + i];
    w += (unsigned long long)(g.rp[x + 1] - g.rp[x]) + 1ull;  // >= |{w in N(x) : w < v}| + 1
  }
  work[v] = w;
}

__global__ __launch_bounds__(256) void rect_acc_kernel(const RectAccParams p) {
  __shared__ WaveLds W[kWavesPerBlock];
  __shared__ int4 s_task;
  __shared__ int s_next, s_ntouched;
  __shared__ int s_wtouched[kWavesPerBlock];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds &L = W[wave];
  const size_t slot_own = (size_t)blockIdx.x * kWavesPerBlock + wave, slot_wg = (size_t)blockIdx.x * kWavesPerBlock;
  unsigned long long cnt = 0;
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long q = atomicAdd(p.queue, 1ull);
      s_task = (q < p.count) ? p.tasks[p.first + q * p.step] : make_int4(-3, -3, -3, -3);
      s_next = 0;
      s_ntouched = 0;
    }
    if (lane == 0) s_wtouched[wave] = 0;
    __syncthreads();
    const int4 t = s_task;
    if (t.x == -3) break;
    const bool heavy = t.y == -2;
    const int v0 = heavy ? t.x : (wave == 0 ? t.x : wave == 1 ? t.y : wave == 2 ? t.z : t.w);
    unsigned *acc = p.acc + (heavy ? slot_wg : slot_own) * p.acc_stride;
    int *touched = p.touched + (heavy ? slot_wg : slot_own) * p.acc_stride;
    int *ntouched = heavy ? &s_ntouched : &s_wtouched[wave];
    // phase 0: the 2-path walk (every increment adds the counter's old value; a vertex whose old value is 0 is listed);
    // phase 1: the listed counters are cleared (one store per touched vertex instead of a second walk over the 2-paths)
    for (int phase = 0; phase < 2; ++phase) {
      if (v0 >= 0) {
        const int r0 = rp[v0];
        const int nitems = (phase == 0) ? p.idx0[v0] : *ntouched;
        int mine = 0;
        for (;;) {
          int bi = 0;
          if (heavy) {
            if (lane == 0) bi = atomicAdd(&s_next, 1);
            bi = readfirst(bi);
          } else {
            bi = mine++;
          }
          if (bi * GM_WAVE >= nitems) break;
          const int i = bi * GM_WAVE + lane;
          if (phase == 0) {
            int llen = 0, kb = 0;
            if (i < nitems) {
              const int x = col[r0 + i];
              kb = rp[x];
              llen = lower_bound(col + kb, rp[x + 1] - kb, v0);  // {w in N(x) : w < v0}
            }
            // (measured: issuing the four returning atomics of a tile group back to back is ~5 % SLOWER than one at a
            // time -- the map updates are bound by the L2 atomic units, not by latency)
            auto inc = [&](const bool *in, const int *key, const int *) {
#pragma unroll
              for (int q = 0; q < kTilesG; ++q) {
                bool first = false;
                if (in[q]) {
                  const unsigned old = __hip_atomic_fetch_add(&acc[key[q]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  cnt += (unsigned long long)old;
                  first = old == 0u;
                }
                const unsigned long long m = __ballot(first);
                if (m) {  // wave-uniform
                  int base = 0;
                  if (lane == 0) base = atomicAdd(ntouched, __popcll(m));
                  base = readfirst(base);
                  if (first) touched[base + rank_below(m)] = key[q];
                }
              }
            };
            flat_pass<SEARCH_NONE>(L, nullptr, col, nullptr, lane, llen, kb, 0, 0, inc);
          } else if (i < nitems) {
            acc[touched[i]] = 0u;
          }
        }
      }
      if (heavy) {  // (t is uniform over the workgroup)
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0;
        __threadfence_block();
        __syncthreads();
      } else {
        __threadfence_block();
        wave_sync();
      }
    }
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64(cnt);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_rect_work(const GraphView &g, const int *idx0, unsigned long long *work, hipStream_t stream) {
  hipLaunchKernelGGL(rect_work_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, idx0, work);
  return hipGetLastError();
}

hipError_t launch_rect_acc(const RectAccParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(rect_acc_kernel, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ---- house by wedge accumulation ---------------------------------------------------------------------------------
// Summing house.h:1-16 over v2 AND exchanging the sums over v1 and v3 (t(a,b) = |N(a) ^ N(b)|, S = N(v0) ^ N(v1)):
//   house = A - B - C - D
//   A = sum_{v0} sum_{v3 != v0} c(v0,v3) * w(v0,v3),   c = |N(v0) ^ N(v3)|,  w = sum_{v1 in N(v0)^N(v3), v1 < v0} t(v0,v1)
//   B = sum_{(v0,v1), v1 < v0} t(v0,v1) * (d(v1) - 1)
//   C = sum_{entries (v0 -> v3)} (t(v0,v3) - 1) * |{v1 in N(v0)^N(v3) : v1 < v0}|
//   D = sum_{(v0,v1), v1 < v0} sum_{w in S} (t(v1,w) - 1)
// (checked against the reference's goldens). A is ONE walk over the 2-paths v0 - v1 - v3 with a vertex-indexed 64-bit map
// (count in the low 24 bits, weighted sum above): a returning atomic add of (delta << 24 | 1), delta = [v1 < v0] t(v0,v1),
// yields the old (c, w) and the pair products telescope to c * w. B, C, D need the per-entry tables t and
// tlt(v0 -> v1) = |{x in S : x < v0}| (edge_tab_kernel) and one more intersection per edge (D). No per-wedge intersections.
__global__ __launch_bounds__(256) void edge_tab_kernel(GraphView g, unsigned *__restrict__ t, unsigned *__restrict__ tlt,
                                                       unsigned long long *__restrict__ queue) {
  __shared__ WaveLds W[kWavesPerBlock];
  __shared__ int AUX[kWavesPerBlock][GM_WAVE];
  const int *__restrict__ rp = g.rp;
  const int *__restrict__ col = g.col;
  const int lane = threadIdx.x & 63;
  WaveLds &L = W[threadIdx.x >> 6];
  int *v0s = AUX[threadIdx.x >> 6];
  const unsigned long long nblk = ((unsigned long long)g.ne + 63ull) / 64ull;
  for (;;) {
    unsigned long long q = 0;
    if (lane == 0) q = atomicAdd(queue, 1ull);
    q = ((unsigned long long)(unsigned)readfirst((int)(q >> 32)) << 32) | (unsigned)readfirst((int)q);
    if (q >= nblk) break;
    const long long e = (long long)q * 64 + lane;
    bool valid = e < (long long)g.ne;
    int v0 = 0, v1 = 0;
    if (valid) {
      int lo = 0, hi = g.nv - 1;  // row of entry e
      while (lo < hi) {
        const int mid = (int)(((long long)lo + hi + 1) >> 1);
        if (rp[mid] <= e) lo = mid; else hi = mid - 1;
      }
      v0 = lo;
      v1 = col[e];
      valid = v1 < v0;  // every undirected edge once; both directed entries are written below
    }
    int llen = 0, kb = 0, sb = 0, sl = 0;
    if (valid) {
      const int r0 = rp[v0], d0 = rp[v0 + 1] - r0, r1 = rp[v1], d1 = rp[v1 + 1] - r1;
      if (d0 <= d1) { llen = d0; kb = r0; sb = r1; sl = d1; } else { llen = d1; kb = r1; sb = r0; sl = d0; }
    }
    L.cnt[lane] = 0u;
    L.qkey[lane] = 0;             // common neighbours below v0
    L.qkey[GM_WAVE + lane] = 0;   // ... below v1
    L.qkey[2 * GM_WAVE + lane] = v1;
    v0s[lane] = v0;
    wave_sync();
    auto act = [&](bool f, int owner, int, int, int, int key) {
      if (!f) return;
      atomicAdd(&L.cnt[owner], 1u);
      if (key < v0s[owner]) atomicAdd(&L.qkey[owner], 1);
      if (key < L.qkey[2 * GM_WAVE + owner]) atomicAdd(&L.qkey[GM_WAVE + owner], 1);
    };
    flat_pass<SEARCH_HBM>(L, nullptr, col, nullptr, lane, llen, kb, sb, sl, act);
    wave_sync();
    if (valid) {
      const unsigned c = L.cnt[lane];
      const long long e10 = (long long)rp[v1] + lower_bound(col + rp[v1], rp[v1 + 1] - rp[v1], v0);
      t[e] = c;
      tlt[e] = (unsigned)L.qkey[lane];
      t[e10] = c;
      tlt[e10] = (unsigned)L.qkey[GM_WAVE + lane];
    }
    wave_sync();
  }
}

__global__ __launch_bounds__(256) void house_work_kernel(GraphView g, unsigned long long *__restrict__ work) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.nv) return;
  unsigned long long w = 0;
  for (int i = g.rp[v]; i < g.rp[v + 1]; ++i) {
    const int x = g.col[i];
    w += (unsigned long long)(g.rp[x + 1] - g.rp[x]) + 1ull;
  }
  work[v] = w;
}

__global__ __launch_bounds__(256) void house_acc_kernel(const HouseAccParams p) {
  __shared__ WaveLds W[kWavesPerBlock];
  __shared__ int AUX[kWavesPerBlock][GM_WAVE];
  __shared__ int4 s_task;
  __shared__ int s_next, s_ntouched;
  __shared__ int s_wtouched[kWavesPerBlock];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds &L = W[wave];
  int *aux = AUX[wave];
  unsigned long long *acc_own = p.acc + ((size_t)blockIdx.x * kWavesPerBlock + wave) * p.acc_stride;
  unsigned long long *acc_wg = p.acc + ((size_t)blockIdx.x * kWavesPerBlock) * p.acc_stride;
  unsigned long long res = 0;  // A - B - C - D of this lane, modulo 2^64 (per centre the total is a count, >= 0)
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long q = atomicAdd(p.queue, 1ull);
      s_task = (q < p.count) ? p.tasks[p.first + q * p.step] : make_int4(-3, -3, -3, -3);
      s_next = 0;
      s_ntouched = 0;
    }
    if (lane == 0) s_wtouched[wave] = 0;
    __syncthreads();
    const int4 tk = s_task;
    if (tk.x == -3) break;
    const bool heavy = tk.y == -2;
    const int v0 = heavy ? tk.x : (wave == 0 ? tk.x : wave == 1 ? tk.y : wave == 2 ? tk.z : tk.w);
    unsigned long long *acc = heavy ? acc_wg : acc_own;
    int *touched = p.touched + ((size_t)blockIdx.x * kWavesPerBlock + (heavy ? 0 : wave)) * p.acc_stride;
    int *ntouched = heavy ? &s_ntouched : &s_wtouched[wave];
    // phases: 0 = B, C, D (tables + one intersection per edge v1 < v0); 1 = the 2-path walk (A), listing the touched
    // vertices; 2 = clear the listed map entries
    for (int phase = 0; phase < 3; ++phase) {
      if (v0 >= 0) {
        const int r0 = rp[v0], d0 = rp[v0 + 1] - r0;
        int mine = 0;
        for (;;) {
          int bi = 0;
          if (heavy) {
            if (lane == 0) bi = atomicAdd(&s_next, 1);
            bi = readfirst(bi);
          } else {
            bi = mine++;
          }
          const int nitems = (phase == 2) ? *ntouched : d0;
          if (bi * GM_WAVE >= nitems) break;
          const int i = bi * GM_WAVE + lane;
          if (phase == 2) {
            if (i < nitems) acc[touched[i]] = 0ull;
            continue;
          }
          const bool valid = i < d0;
          const int v1 = valid ? col[r0 + i] : 0;
          const int r1 = rp[v1], d1 = rp[v1 + 1] - r1;
          const unsigned te = valid ? p.t[r0 + i] : 0u;
          if (phase == 0) {
            const bool sb = valid && v1 < v0;
            if (valid) res -= (unsigned long long)p.tlt[r0 + i] * (unsigned long long)(te - 1u + (te == 0u ? 1u : 0u));  // C (t = 0 => tlt = 0)
            if (sb) res -= (unsigned long long)te * (unsigned long long)(d1 - 1);                                         // B
            if (sb) res += (unsigned long long)te;                                                                        // the "- 1" of D
            int llen = 0, kb = 0, sbase = 0, sl = 0;
            if (sb && te > 0u) {
              if (d1 <= d0) { llen = d1; kb = r1; sbase = r0; sl = d0 | (1 << 30); }  // keys from N(v1): flag 1, kidx = position in N(v1)
              else { llen = d0; kb = r0; sbase = r1; sl = d1; }                       // keys from N(v0): pos = position in N(v1)
            }
            aux[lane] = r1;
            wave_sync();
            unsigned long long dsum = 0;
            auto act = [&](bool f, int owner, int kidx, int pos, int flag, int) {
              if (f) dsum += (unsigned long long)p.t[aux[owner] + (flag ? kidx : pos)];  // t(v1, w)
            };
            flat_pass<SEARCH_HBM>(L, nullptr, col, nullptr, lane, llen, kb, sbase, sl, act);
            res -= dsum;  // D
            wave_sync();
          } else {
            L.cnt[lane] = (valid && v1 < v0) ? te : 0u;  // delta of this entry's 2-paths
            wave_sync();
            const int llen = valid ? d1 : 0;
            if (phase == 1) {
              auto inc = [&](const bool *in, const int *key, const int *own) {
#pragma unroll
                for (int q = 0; q < kTilesG; ++q) {
                  bool first = false;
                  if (in[q] && key[q] != v0) {
                    const unsigned long long delta = (unsigned long long)L.cnt[own[q] - 1];
                    const unsigned long long old = __hip_atomic_fetch_add(&acc[key[q]], (delta << 24) | 1ull, __ATOMIC_RELAXED,
                                                                          __HIP_MEMORY_SCOPE_WORKGROUP);
                    res += delta * (old & 0xffffffull) + (old >> 24) + delta;  // A
                    first = old == 0ull;
                  }
                  const unsigned long long m = __ballot(first);
                  if (m) {  // wave-uniform: list the newly touched vertices
                    int base = 0;
                    if (lane == 0) base = atomicAdd(ntouched, __popcll(m));
                    base = readfirst(base);
                    if (first) touched[base + rank_below(m)] = key[q];
                  }
                }
              };
              flat_pass<SEARCH_NONE>(L, nullptr, col, nullptr, lane, llen, r1, 0, 0, inc);
            }
            wave_sync();
          }
        }
      }
      if (heavy) {
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0;
        __threadfence_block();
        __syncthreads();
      } else {
        __threadfence_block();
        wave_sync();
      }
    }
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64(res);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_edge_tab(const GraphView &g, unsigned *t, unsigned *tlt, unsigned long long *queue, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(edge_tab_kernel, dim3((unsigned)grid_blocks), dim3(256), 0, stream, g, t, tlt, queue);
  return hipGetLastError();
}

hipError_t launch_house_work(const GraphView &g, unsigned long long *work, hipStream_t stream) {
  hipLaunchKernelGGL(house_work_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, work);
  return hipGetLastError();
}

hipError_t launch_house_acc(const HouseAccParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(house_acc_kernel, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ---- pentagon by wedge accumulation --------------------------------------------------------------------------------
// pentagon.h:2-17 counts every 5-cycle once, at its largest vertex v0. With M = {b in N(v0) : b < v0} and the rectangle
// kernel's map cm[x] = |N(x) ^ M| (x < v0), the closed walks v0 - a - x - y - b - v0 (a, b in M; x ~ y; x, y < v0) number
//   W(v0) = sum_{x < v0} cm[x] * sum_{y in N(x), y < v0} cm[y];
// a 5-cycle is two of them, the rest repeat a vertex (a = b: a triangle under v0; a = y or x = b: an edge inside M):
//   2 pentagon = sum_{v0} [ W - 2 N2 + N23 ] - N1,    N2 = sum_{a in M} |{x in N(a) : x < v0}| * |N(a) ^ M|,   N23 = sum_{a in M} |N(a) ^ M|,
//   N1 = 2 sum_{edges x < y} [ tlo (above(x,y) + d(y) - idx0(y)) + tmid above(x,y) ],   above(x,y) = |{z in N(x) : z > y}|,
// with tlo / tmid = common neighbours of x, y below x / between them (both from the tlt table: tlt(x->y) and tlt(y->x) - tlt(x->y)),
// |N(a) ^ M| = tlt(v0->a). (Checked against the reference's goldens.) Per centre: one 2-path walk that fills the map and
// lists the touched vertices, one walk over the rows of the touched vertices, a row of table arithmetic, one clearing pass --
// no intersections at all. The N1 terms are grouped by the row of the edge's larger endpoint, so a centre's contribution is an
// even, possibly negative integer: the 64-bit sum is kept in two's complement and halved (arithmetically) at the end.
__global__ __launch_bounds__(256) void pent_acc_kernel(const PentAccParams p) {
  __shared__ WaveLds W[kWavesPerBlock];
  __shared__ int4 s_task;
  __shared__ int s_next, s_ntouched;
  __shared__ int s_wtouched[kWavesPerBlock];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds &L = W[wave];
  const size_t slot_own = (size_t)blockIdx.x * kWavesPerBlock + wave, slot_wg = (size_t)blockIdx.x * kWavesPerBlock;
  long long res = 0;
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long q = atomicAdd(p.queue, 1ull);
      s_task = (q < p.count) ? p.tasks[p.first + q * p.step] : make_int4(-3, -3, -3, -3);
      s_next = 0;
      s_ntouched = 0;
    }
    if (lane == 0) s_wtouched[wave] = 0;
    __syncthreads();
    const int4 tk = s_task;
    if (tk.x == -3) break;
    const bool heavy = tk.y == -2;
    const int v0 = heavy ? tk.x : (wave == 0 ? tk.x : wave == 1 ? tk.y : wave == 2 ? tk.z : tk.w);
    unsigned *acc = p.acc + (heavy ? slot_wg : slot_own) * p.acc_stride;
    int *touched = p.touched + (heavy ? slot_wg : slot_own) * p.acc_stride;
    int *ntouched = heavy ? &s_ntouched : &s_wtouched[wave];
    // phases: 0 = fill the map over the 2-paths v0 - b - x and list the touched x; 1 = W over the touched rows and the
    // table terms of the entries (v0 -> a), a < v0; 2 = clear the map
    for (int phase = 0; phase < 3; ++phase) {
      if (v0 >= 0) {
        const int r0 = rp[v0], n0 = p.idx0[v0], d0 = rp[v0 + 1] - r0;
        const int nt = (phase == 0) ? 0 : *ntouched;          // (written in phase 0, barrier / fence in between)
        const int nitems = (phase == 0) ? n0 : nt;            // entries b in M  /  touched vertices
        int mine = 0;
        for (;;) {
          int bi = 0;
          if (heavy) {
            if (lane == 0) bi = atomicAdd(&s_next, 1);
            bi = readfirst(bi);
          } else {
            bi = mine++;
          }
          if (bi * GM_WAVE >= nitems) break;
          const int i = bi * GM_WAVE + lane;
          const bool valid = i < nitems;
          if (phase == 0) {
            int llen = 0, kb = 0;
            if (valid) {
              const int b = col[r0 + i];
              kb = rp[b];
              llen = lower_bound(col + kb, rp[b + 1] - kb, v0);
            }
            auto inc = [&](const bool *in, const int *key, const int *) {
#pragma unroll
              for (int q = 0; q < kTilesG; ++q) {
                bool first = false;
                if (in[q]) first = __hip_atomic_fetch_add(&acc[key[q]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u;
                const unsigned long long m = __ballot(first);
                if (m) {  // wave-uniform: append the newly touched vertices
                  int base = 0;
                  if (lane == 0) base = atomicAdd(ntouched, __popcll(m));
                  base = readfirst(base);
                  if (first) touched[base + rank_below(m)] = key[q];
                }
              }
            };
            flat_pass<SEARCH_NONE>(L, nullptr, col, nullptr, lane, llen, kb, 0, 0, inc);
          } else {
            const int x = valid ? touched[i] : 0;
            if (phase == 1) {
              const int rx = rp[x];
              const int llen = valid ? lower_bound(col + rx, rp[x + 1] - rx, v0) : 0;
              L.cnt[lane] = valid ? __hip_atomic_load(&acc[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;  // cm[x] (not through L1)
              wave_sync();
              auto walk = [&](const bool *in, const int *key, const int *own) {
#pragma unroll
                for (int q = 0; q < kTilesG; ++q)
                  if (in[q]) {
                    const unsigned cy = __hip_atomic_load(&acc[key[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    res += (long long)((unsigned long long)L.cnt[own[q] - 1] * (unsigned long long)cy);  // W
                  }
              };
              flat_pass<SEARCH_NONE>(L, nullptr, col, nullptr, lane, llen, rx, 0, 0, walk);
              wave_sync();
            } else if (valid) {
              acc[x] = 0u;
            }
          }
        }
        if (phase == 1) {  // table terms of the entries (v0 -> a), a < v0 (lanes / waves stride them; no flat pass needed)
          const int wstep = heavy ? kWavesPerBlock * GM_WAVE : GM_WAVE, wfirst = heavy ? (int)threadIdx.x : lane;
          for (int i = wfirst; i < n0; i += wstep) {
            const int e = r0 + i, a = col[e];
            const int ra = rp[a], da = rp[a + 1] - ra;
            const int pos = lower_bound(col + ra, da, v0);  // = |{x in N(a) : x < v0}|, and the entry (a -> v0) is ra + pos
            const long long tl = (long long)p.tlt[e];       // common neighbours below v0 = |N(a) ^ M| = tlo + tmid
            const long long tlo = (long long)p.tlt[ra + pos];
            const long long above = (long long)(da - pos - 1);
            res += tl - 2ll * (long long)pos * tl;                                                   // N23 - 2 N2
            res -= 2ll * (tlo * (above + (long long)(d0 - n0)) + (tl - tlo) * above);             // N1, row v0
          }
        }
      }
      if (heavy) {
        __syncthreads();
        if (threadIdx.x == 0) s_next = 0;
        __threadfence_block();
        __syncthreads();
      } else {
        __threadfence_block();
        wave_sync();
      }
    }
    __syncthreads();
  }
  const unsigned long long s0 = wave_sum_u64((unsigned long long)res);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_pent_acc(const PentAccParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(pent_acc_kernel, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// ---- house, flattened ----------------------------------------------------------------------------------------
// src/sgl/cpu_kernels/house.h:1-16:  for v0, v1 in N(v0) (v1 < v0), S = N(v0) ^ N(v1), v2 in S, v3 in N(v1) \ {v0, v2}:
//                                        count += |N(v0) ^ N(v3) \ {v1, v2}|
// Summing over v2 first (v1 is always common to N(v0) and N(v3); v2 in S is common iff v2 in N(v3)):
//   house(v0,v1) = sum_{v3 in N(v1), v3 != v0} [ (|S| - [v3 in S]) * (|N(v0) ^ N(v3)| - 1) - |S ^ N(v3)| ]
// (checked against the reference's goldens). Tasks = (v0, v1, v3): a wave takes one entry (v0 -> v1) and 64 of its v3;
// |S| is one cooperative intersection per wave, the 64 intersections N(v0) ^ N(v3) run as ONE flattened pass whose
// match handler also tests the matched key against N(v1) (that is |S ^ N(v3)|).
__global__ __launch_bounds__(256) void house_blocks_kernel(GraphView g, unsigned *__restrict__ nblk) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.ne) return;
  int lo = 0, hi = g.nv - 1;  // row of entry e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (g.rp[mid] <= e) lo = mid; else hi = mid - 1;
  }
  const int v1 = g.col[e];
  nblk[e] = (v1 < lo) ? (unsigned)((g.rp[v1 + 1] - g.rp[v1] + 63) / 64) : 0u;
}

constexpr int kHouseBitWords = 512;  // per-wave LDS bitmap over the positions of N(v0): which of them are in S

__global__ __launch_bounds__(256) void house_flat_kernel(const HouseParams p) {
  __shared__ WaveLds W[kWavesPerBlock];
  __shared__ unsigned SB[kWavesPerBlock][kHouseBitWords];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63;
  WaveLds &L = W[threadIdx.x >> 6];
  unsigned *sbits = SB[threadIdx.x >> 6];
  unsigned long long cnt = 0;
  for (;;) {
    unsigned long long q = 0;
    if (lane == 0) q = atomicAdd(p.queue, 1ull);
    q = ((unsigned long long)(unsigned)readfirst((int)(q >> 32)) << 32) | (unsigned)readfirst((int)q);
    if (q >= p.count) break;
    const unsigned long long gid = p.first + q * p.step;
    const unsigned long long b0 = gid * (unsigned long long)p.group;
    const unsigned long long b1 = min(p.nblocks, b0 + (unsigned long long)p.group);
    int elo = 0, ehi = p.g.ne - 1;  // entry of the first block: largest e with entry_prefix[e] <= b0
    while (elo < ehi) {
      const int mid = (int)(((long long)elo + ehi + 1) >> 1);
      if (p.entry_prefix[mid] <= b0) elo = mid; else ehi = mid - 1;
    }
    int e = elo;
    int vlo = 0, vhi = p.g.nv - 1;  // row of that entry
    while (vlo < vhi) {
      const int mid = (vlo + vhi + 1) >> 1;
      if (rp[mid] <= e) vlo = mid; else vhi = mid - 1;
    }
    int v0 = vlo;
    int cached_e = -1;
    long long sn = 0;
    bool use_bits = false;
    for (unsigned long long blk = b0; blk < b1; ++blk) {
      while (p.entry_prefix[e + 1] <= blk) ++e;
      while (rp[v0 + 1] <= e) ++v0;
      const int v1 = col[e];
      const int r0 = rp[v0], d0 = rp[v0 + 1] - r0;
      const int r1 = rp[v1], d1 = rp[v1 + 1] - r1;
      if (e != cached_e) {  // S = N(v0) ^ N(v1): its size, and (rows up to 16 K entries) its members as bits over N(v0)
        cached_e = e;
        use_bits = (d0 <= kHouseBitWords * 32) & !p.no_bits;
        if (use_bits) {
          for (int i = lane; i < (d0 + 31) / 32; i += GM_WAVE) sbits[i] = 0u;
          wave_sync();
        }
        unsigned c = 0;
        if (d0 <= d1) {
          for (int i = lane; i < d0; i += GM_WAVE) {
            int pos;
            if (contains(col + r1, d1, col[r0 + i], &pos)) {
              ++c;
              if (use_bits) atomicOr(&sbits[i >> 5], 1u << (i & 31));
            }
          }
        } else {
          for (int i = lane; i < d1; i += GM_WAVE) {
            int pos;
            if (contains(col + r0, d0, col[r1 + i], &pos)) {
              ++c;
              if (use_bits) atomicOr(&sbits[pos >> 5], 1u << (pos & 31));
            }
          }
        }
        sn = (long long)wave_sum((int)c);
        wave_sync();
      }
      const int k = (int)(blk - p.entry_prefix[e]) * 64 + lane;
      bool valid = k < d1;
      int v3 = valid ? col[r1 + k] : 0;
      valid = valid && v3 != v0;
      int llen = 0, key_base = 0, s_base = 0, s_len = 0;
      long long in_s = 0;
      if (valid) {
        int pos;
        in_s = contains(col + r0, d0, v3, &pos) ? 1 : 0;  // v3 in N(v1) already, so v3 in S  <=>  v3 in N(v0)
        const int r3 = rp[v3], d3 = rp[v3 + 1] - r3;
        if (d0 <= d3) { llen = d0; key_base = r0; s_base = r3; s_len = d3; }             // keys from N(v0): flag 0
        else { llen = d3; key_base = r3; s_base = r0; s_len = d0 | (1 << 30); }          // keys from N(v3): flag 1
      }
      L.cnt[lane] = 0u;   // |N(v0) ^ N(v3)| of this lane's task
      L.qkey[lane] = 0;   // |S ^ N(v3)|
      wave_sync();
      auto act = [&](bool f, int owner, int kidx, int pos, int flag, int key) {
        if (!f) return;
        atomicAdd(&L.cnt[owner], 1u);
        bool ins;
        if (use_bits) {
          const int p0 = flag ? pos : kidx;  // position of the matched key inside N(v0)
          ins = (sbits[p0 >> 5] >> (p0 & 31)) & 1u;
        } else {
          int pp;
          ins = contains(col + r1, d1, key, &pp);
        }
        if (ins) atomicAdd(&L.qkey[owner], 1);
      };
      flat_pass<SEARCH_HBM>(L, nullptr, col, nullptr, lane, llen, key_base, s_base, s_len, act);
      wave_sync();
      if (valid) {
        const long long ca = (long long)L.cnt[lane], cs = (long long)L.qkey[lane];
        cnt += (unsigned long long)((sn - in_s) * (ca - 1) - cs);
      }
      wave_sync();
    }
  }
  const unsigned long long s0 = wave_sum_u64(cnt);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_house_blocks(const GraphView &g, unsigned *nblk, hipStream_t stream) {
  hipLaunchKernelGGL(house_blocks_kernel, dim3((unsigned)((g.ne + 255) / 256)), dim3(256), 0, stream, g, nblk);
  return hipGetLastError();
}

hipError_t launch_house_flat(const HouseParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(house_flat_kernel, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_idx0(const GraphView &g, int *idx0, hipStream_t stream) {
  hipLaunchKernelGGL(idx0_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, idx0);
  return hipGetLastError();
}

hipError_t launch_rect_flat(const RectParams &p, bool pentagon, int grid_blocks, hipStream_t stream) {
  if (pentagon) hipLaunchKernelGGL(rect_flat_kernel<true>, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(rect_flat_kernel<false>, dim3((unsigned)grid_blocks), dim3(256), 0, stream, p);
  return hipGetLastError();
}

size_t mine_lds_bytes(Pattern pat) {
  switch (pat) {
    case PAT_CLIQUE4:
    case PAT_CLIQUEK: return sizeof(BlockLds<PAT_CLIQUE4>);
    case PAT_DIAMOND:
    case PAT_MOTIF3:
    case PAT_MOTIF4E: return sizeof(BlockLds<PAT_DIAMOND>);
    default: return sizeof(BlockLds<PAT_TC>);
  }
}

hipError_t launch_mine(Pattern pat, const MineParams &p, int grid_blocks, hipStream_t stream) {
  dim3 grid((unsigned)grid_blocks), block(kWavesPerBlock * GM_WAVE);
  switch (pat) {
    case PAT_TC: hipLaunchKernelGGL(mine_kernel<PAT_TC>, grid, block, 0, stream, p); break;
    case PAT_DIAMOND: hipLaunchKernelGGL(mine_kernel<PAT_DIAMOND>, grid, block, 0, stream, p); break;
    case PAT_MOTIF4E: hipLaunchKernelGGL(mine_kernel<PAT_MOTIF4E>, grid, block, 0, stream, p); break;
    case PAT_DAGSTATS: hipLaunchKernelGGL(mine_kernel<PAT_DAGSTATS>, grid, block, 0, stream, p); break;
    case PAT_MOTIF3: hipLaunchKernelGGL(mine_kernel<PAT_MOTIF3>, grid, block, 0, stream, p); break;
    case PAT_CLIQUE4: hipLaunchKernelGGL(mine_kernel<PAT_CLIQUE4>, grid, block, 0, stream, p); break;
    case PAT_CLIQUEK: hipLaunchKernelGGL(mine_kernel<PAT_CLIQUEK>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gm
