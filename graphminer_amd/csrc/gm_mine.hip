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
#include "gm_chunk.h"

namespace gm {

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
  __shared__ int s_next, s_ntouched, s_cut;
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
      s_cut = (q < p.count && p.first + q * p.step < p.n_cut) ? 1 : 0;
      s_next = 0;
      s_ntouched = 0;
    }
    if (lane == 0) s_wtouched[wave] = 0;
    __syncthreads();
    const int4 t = s_task;
    if (t.x == -3) break;
    const bool use_cut = s_cut != 0;  // (the ends at or above p.cut of this task's centres belong to rect_lds_kernel)
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
              if (use_cut && p.cut <= v0) llen = p.bnd0[(size_t)x * (size_t)p.bnd_stride] - kb;  // {w in N(x) : w < cut}
              else llen = lower_bound(col + kb, rp[x + 1] - kb, v0);  // {w in N(x) : w < v0}
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

// ---- rectangle: the counter maps of the heavy centres in LDS (round 6) -----------------------------------------------------------------
// rect_acc_kernel's maps are vertex-indexed arrays in global memory, one per wave, and every 2-path is a returning atomic that goes to L2
// and beyond (R-MAT-20 ef 16: 5.6 G 2-paths, 42 bytes of traffic each, 344 ms -- 0.09 of the HBM rate, bound by the atomic units).  On a
// graph numbered ascending in degree the ends w of the 2-paths are the hubs: 61 % lie in the last 32 K ids, 96 % in the last 256 K.  So
// the id space [cut, nv) is cut into ranges (32 K ids of 32-bit counters at the hubs; 64 K / 128 K ids of 16- / 8-bit counters packed into
// words where no vertex of the range has a degree of 2^16 / 2^8 -- a counter c(w) never exceeds d(w)) and a workgroup takes (centre v0, range k): it walks the neighbours x of v0
// below v0, the entries of row x inside the range come from a table of row bounds (bnd: one binary search per (x, range boundary), once per
// graph), every 2-path is ONE non-returning LDS atomic into a dense map, and sum C(c, 2) is taken when the map is read back and cleared.
// The ends below cut stay with rect_acc_kernel (RectAccParams::cut).  Same count: rectangle.h:1-11 as restated above rect_work_kernel.
#ifndef GM_RECT_LDS_TILES
#define GM_RECT_LDS_TILES 8
#endif
constexpr int kRectLdsTiles = GM_RECT_LDS_TILES;  // tiles of 64 keys requested together by a wave of rect_lds_kernel
struct alignas(16) WaveLdsFlat {  // what flat_pass<SEARCH_NONE> uses of the per-wave scratch
  int4 desc[GM_WAVE];
  unsigned char marks[kMarkWindow];
};
// every key of the lists (kb, llen) of the 64 lanes, flattened into tiles of 64 -- flat_pass<SEARCH_NONE> with the NEXT group of TG tiles
// requested before the current one is handed to f(in[], key[]): the walk of rect_lds_kernel is one global load per key followed by one
// LDS atomic, and with four loads per lane in flight and nothing behind them a 16-wave workgroup moved 1.2 G keys/s per CU (17.2 ms for the
// 5.6 G 2-paths of R-MAT-20 ef 16), the latency of the loads, not the LDS.
template <int TG, class LT, class F>
__device__ __forceinline__ void flat_walk(LT &L, const int *__restrict__ col, const int lane, const int llen, const int kb, F f) {
  static_assert(kMarkWindow % (GM_WAVE * TG) == 0, "a window of owner marks is a whole number of tile groups");
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  if (total == 0) return;  // wave-uniform
  const int off = incl - llen;
  L.desc[lane] = make_int4(kb, off, 0, 0);
  unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
  int carry = 0;
  int key[TG];
  bool in[TG], have = false;
  for (int wb = 0; wb < total; wb += kMarkWindow) {
    const int wn = min(kMarkWindow, total - wb);
    const int nwords = ((wn + GM_WAVE * TG - 1) / (GM_WAVE * TG)) * (GM_WAVE * TG / 4);
    for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
    wave_sync();
    if (llen > 0 && off >= wb && off < wb + kMarkWindow) L.marks[off - wb] = (unsigned char)(lane + 1);
    wave_sync();
    for (int t = 0; t < wn; t += GM_WAVE * TG) {
      int own[TG], nkey[TG];
      bool nin[TG];
#pragma unroll
      for (int q = 0; q < TG; ++q) own[q] = (int)L.marks[t + q * GM_WAVE + lane];
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        own[q] = max(wave_incl_scan_max(own[q]), carry);
        carry = readlane(own[q], GM_WAVE - 1);
      }
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int pp = wb + t + q * GM_WAVE + lane;
        nin[q] = pp < total;
        const int4 d = L.desc[nin[q] ? own[q] - 1 : 0];
        nkey[q] = col[nin[q] ? d.x + (pp - d.y) : 0];  // unconditional load (select on the index)
      }
      if (have) f(in, key);  // (the previous group, while this one's keys are on their way)
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        key[q] = nkey[q];
        in[q] = nin[q];
      }
      have = true;
    }
    wave_sync();
  }
  if (have) f(in, key);
}

struct alignas(16) RectLds {
  unsigned map[kRectLdsWords];
  WaveLdsFlat w[kRectLdsWaves];
  int2 task;
  int next;
  unsigned any[2];  // (by task parity: the flag of the next task is cleared while this one's map is read back)
};

__global__ __launch_bounds__(256) void rect_bounds_kernel(GraphView g, RectLdsRanges rr, int *__restrict__ bnd) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.nv) return;
  const int r = g.rp[x], d = g.rp[x + 1] - r;
  int at = 0;
  for (int k = 0; k <= rr.n; ++k) {
    at += (rr.rb[k] >= g.nv) ? d - at : lower_bound(g.col + r + at, d - at, rr.rb[k]);
    bnd[(size_t)x * (size_t)(rr.n + 1) + (size_t)k] = r + at;
  }
}

// largest degree per block of kRectLdsWords ids, counted from the LAST id down (the ranges are whole blocks)
__global__ __launch_bounds__(256) void rect_blockmax_kernel(GraphView g, int *__restrict__ blockmax) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int xc = min(x, g.nv - 1);
  const int b = (g.nv - 1 - xc) / kRectLdsWords, d = x < g.nv ? g.rp[xc + 1] - g.rp[xc] : 0;
  // (a wave's 64 ids share a block except where one ends: one atomic per wave there, same-address atomics are served one after the other)
  const int b0 = readfirst(b);
  const bool same = __ballot(b != b0) == 0ull;
  const int m = wave_max_nonneg(d);
  if (same) {
    if ((threadIdx.x & 63) == 0) atomicMax(&blockmax[b0], m);
  } else {
    atomicMax(&blockmax[b], d);
  }
}

// 2-paths of centre v whose end lies below cut (the part rect_acc_kernel keeps for a centre whose other ends go to the LDS maps)
__global__ __launch_bounds__(256) void rect_work_cut_kernel(GraphView g, const int *__restrict__ idx0, const int *__restrict__ bnd0, int bnd_stride,
                                                            unsigned long long *__restrict__ work) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.nv) return;
  unsigned long long w = 0, wc = 0;
  const int r0 = g.rp[v];
  for (int i = 0; i < idx0[v]; ++i) {
    const int x = g.col[r0 + i];
    w += (unsigned long long)(g.rp[x + 1] - g.rp[x]) + 1ull;  // (as rect_work_kernel)
    wc += (unsigned long long)(bnd0[(size_t)x * (size_t)bnd_stride] - g.rp[x]);
  }
  work[v] = w;
  work[(size_t)g.nv + (size_t)v] = wc;
}

#ifndef GM_RECT_LDS_LONG
#define GM_RECT_LDS_LONG 64
#endif
constexpr int kRectLdsLong = GM_RECT_LDS_LONG;  // keys of a row inside the range from which the wave takes the row on its own

__global__ __launch_bounds__(kRectLdsWaves *GM_WAVE) void rect_lds_kernel(const RectLdsParams p) {
  __shared__ RectLds S;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  constexpr int nthreads = kRectLdsWaves * GM_WAVE;
  WaveLdsFlat &L = S.w[wave];
  for (int i = tid; i < kRectLdsWords; i += nthreads) S.map[i] = 0u;
  unsigned long long cnt = 0;
  const int bs = p.r.n + 1;  // bounds per row
  auto fetch = [&]() {
    const unsigned long long q = atomicAdd(p.queue, 1ull);
    return (q < p.count) ? p.tasks[p.first + q * p.step] : make_int2(-3, -3);
  };
  if (tid == 0) {
    S.task = fetch();
    S.next = 0;
    S.any[0] = S.any[1] = 0u;
  }
  __syncthreads();
  int ph = 0;  // ranges this workgroup has walked (the parity of the "any key" flag)
  // the keys (kb, llen) of this wave's 64 rows into the map of range k; keys >= vlim are left out
  // RTN: the atomic returns the counter and the old value is summed on the spot (0 + 1 + .. + (c - 1) = C(c, 2), as rect_acc_kernel does):
  // the map is then cleared blindly instead of read back -- for the centres whose ranges hold few keys (measured: reading 128 KB of counters
  // back after every one of the 190 K (centre, range) walks was 4.2 of the kernel's 12.5 ms)
  auto walk = [&](const int kb, int llen, const int k, const int vlim, const bool RTN) {
    const int lo = p.r.rb[k], lb = p.r.lb[k], sh = 5 - lb;  // counters of 2^lb bits, 2^sh of them in a word
    const unsigned fm = (1u << sh) - 1u;
    const unsigned cm = lb == 5 ? 0xffffffffu : (1u << (1 << lb)) - 1u;
    auto add = [&](const int key) {
      const unsigned off = (unsigned)(key - lo);
      const unsigned fs = (off & fm) << lb;
      if (key < vlim) {
        if (RTN) cnt += (unsigned long long)((atomicAdd(&S.map[off >> sh], 1u << fs) >> fs) & cm);
        else atomicAdd(&S.map[off >> sh], 1u << fs);  // (result unused: ds_add_u32; a field never carries: c(w) <= d(w) < 2^(2^lb))
      }
    };
    // a row with kRectLdsLong keys or more inside the range (a hub centre's neighbours in the hubs' range): the lanes stride it, no flattening
    unsigned long long lm = __ballot(llen >= kRectLdsLong);
    while (lm) {  // wave-uniform
      const int l = __ffsll((long long)lm) - 1;
      lm &= lm - 1ull;
      const int base = readlane(kb, l), len = readlane(llen, l);
      for (int j0 = 0; j0 < len; j0 += 4 * GM_WAVE) {
        int key[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) key[q] = col[base + min(j0 + q * GM_WAVE + lane, len - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j0 + q * GM_WAVE + lane < len) add(key[q]);
      }
    }
    if (llen >= kRectLdsLong) llen = 0;
    auto inc = [&](const bool *in, const int *key) {
#pragma unroll
      for (int q = 0; q < kRectLdsTiles; ++q)
        if (in[q]) add(key[q]);
    };
#if !defined(GM_RECT_ABL) || GM_RECT_ABL != 1
    flat_walk<kRectLdsTiles>(L, col, lane, llen, kb, inc);
#endif
  };
  // every wave is done with range k of centre v0: sum C(c, 2) over the map, cleared for the next range.  `nt`: the task to publish (the
  // last thread holds it), null = keep the current one
  auto finish = [&](const unsigned any, const int k, const int v0, const int2 *nt, const bool RTN) {
    if (__ballot(any != 0u) != 0ull && lane == 0) S.any[ph & 1] = 1u;
    __syncthreads();
    if (tid == nthreads - 1) {
      if (nt) S.task = *nt;
      S.next = 0;
      S.any[(ph + 1) & 1] = 0u;
    }
#if defined(GM_RECT_ABL) && GM_RECT_ABL == 2
    if (false) {
#else
    if (S.any[ph & 1]) {  // (workgroup-uniform)
#endif
      const int lo = p.r.rb[k], lb = p.r.lb[k], sh = 5 - lb;
      const unsigned fm = (1u << sh) - 1u;
      const int words = (min(p.r.rb[k + 1], v0) - lo + (int)fm) >> sh;
      const unsigned cm = lb == 5 ? 0xffffffffu : (1u << (1 << lb)) - 1u;
      uint4 *m4 = reinterpret_cast<uint4 *>(S.map);
      if (RTN) {
        for (int i = tid; i < ((words + 3) >> 2); i += nthreads) m4[i] = make_uint4(0u, 0u, 0u, 0u);
      } else
      for (int i = tid; i < ((words + 3) >> 2); i += nthreads) {  // (the words behind the last one of the range stay zero)
        const uint4 v4 = m4[i];
        if ((v4.x | v4.y | v4.z | v4.w) == 0u) continue;
        m4[i] = make_uint4(0u, 0u, 0u, 0u);
        const unsigned vv[4] = {v4.x, v4.y, v4.z, v4.w};
        if (lb == 5) {
#pragma unroll
          for (int u = 0; u < 4; ++u) cnt += (unsigned long long)vv[u] * (unsigned long long)(vv[u] - (vv[u] ? 1u : 0u)) / 2ull;
        } else {  // packed counters are below 2^16: 32-bit arithmetic, the sum of a word's fields too
          unsigned s32 = 0u;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            for (unsigned f = 0; f <= fm; ++f) {
              const unsigned c = (vv[u] >> (f << lb)) & cm;
              s32 += (c * (c - (c ? 1u : 0u))) >> 1;
            }
          cnt += (unsigned long long)s32;
        }
      }
    }
    __syncthreads();
    ++ph;
  };
  for (;;) {
    const int2 t = S.task;
    if (t.x == -3) break;
    // the NEXT task is dequeued now and published after this one's first walk: a task used to begin with the round trip of the dequeue, its
    // record, the neighbours of its centre and their bounds, one after the other, with the whole CU waiting (one workgroup fits its LDS)
    int2 nt = make_int2(-3, -3);
    if (tid == nthreads - 1) nt = fetch();
    const int v0 = t.x;
    const int r0 = rp[v0], nitems = p.idx0[v0];
    if (t.y >= 0) {  // ---- ONE range of a centre with more neighbours than the workgroup has threads: batches of 64 rows, dealt to the waves
      const int k = t.y;
      const bool clip = p.r.rb[k + 1] > v0;  // the range holds v0: rows are cut at v0, not at the next boundary
      unsigned any = 0u;
      for (;;) {
        int bi = 0;
        if (lane == 0) bi = atomicAdd(&S.next, 1);
        bi = readfirst(bi);
        if (bi * GM_WAVE >= nitems) break;
        const int i = bi * GM_WAVE + lane;
        int llen = 0, kb = 0;
        if (i < nitems) {
          const int x = col[r0 + i];
          const int *__restrict__ bx = p.bnd + (size_t)x * (size_t)bs + (size_t)k;
          kb = bx[0];
          const int ke = bx[1];
          llen = clip ? lower_bound(col + kb, ke - kb, v0) : ke - kb;
        }
        any |= (unsigned)llen;
        walk(kb, llen, k, 0x7fffffff, false);
      }
      finish(any, k, v0, &nt, false);
    } else {  // ---- EVERY range of a centre with at most one neighbour per thread, from its own range down: a thread keeps its row, the end of
              // one range is the start of the one above it, and the next start is requested while this range is walked
      int ktop = 0;
      while (ktop + 1 < p.r.n && p.r.rb[ktop + 1] < v0) ++ktop;  // the range that holds v0 - 1
      // (row = lane * waves + wave: a centre with 200 neighbours gives every wave a dozen rows, not four waves 64 each and twelve none --
      // the keys of a wave's rows are flattened over its lanes anyway)
      const int row = lane * kRectLdsWaves + wave;
      const bool valid = row < nitems;
      const int x = valid ? col[r0 + row] : 0;
      const int *__restrict__ bx = p.bnd + (size_t)x * (size_t)bs;
      int kb = valid ? bx[ktop] : 0, ke = valid ? bx[ktop + 1] : 0;  // (the top range is walked to its end: the keys >= v0 are dropped by value)
      for (int k = ktop; k >= 0; --k) {
        const int nkb = (valid && k > 0) ? bx[k - 1] : 0;
        const int llen = ke - kb;
        walk(kb, llen, k, k == ktop ? v0 : 0x7fffffff, true);
        finish((unsigned)llen, k, v0, k == ktop ? &nt : nullptr, true);
        ke = kb;
        kb = nkb;
      }
    }
  }
  const unsigned long long s0 = wave_sum_u64(cnt);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_rect_lds(const RectLdsParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(rect_lds_kernel, dim3((unsigned)grid_blocks), dim3(kRectLdsWaves * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_rect_bounds(const GraphView &g, const RectLdsRanges &r, int *bnd, hipStream_t stream) {
  hipLaunchKernelGGL(rect_bounds_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, r, bnd);
  return hipGetLastError();
}
hipError_t launch_rect_blockmax(const GraphView &g, int *blockmax, hipStream_t stream) {
  hipLaunchKernelGGL(rect_blockmax_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, blockmax);
  return hipGetLastError();
}
hipError_t launch_rect_work_cut(const GraphView &g, const int *idx0, const int *bnd0, int bnd_stride, unsigned long long *work, hipStream_t stream) {
  hipLaunchKernelGGL(rect_work_cut_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, idx0, bnd0, bnd_stride, work);
  return hipGetLastError();
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
  __shared__ int4 s_task;
  __shared__ int s_next, s_ntouched, s_cut;
  __shared__ int s_wtouched[kWavesPerBlock];
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveLds &L = W[wave];
  unsigned long long *acc_own = p.acc + ((size_t)blockIdx.x * kWavesPerBlock + wave) * p.acc_stride;
  unsigned long long *acc_wg = p.acc + ((size_t)blockIdx.x * kWavesPerBlock) * p.acc_stride;
  unsigned long long res = 0;  // A - B - C - D of this lane, modulo 2^64 (per centre the total is a count, >= 0)
  for (;;) {
    if (threadIdx.x == 0) {
      const unsigned long long q = atomicAdd(p.queue, 1ull);
      s_task = (q < p.count) ? p.tasks[p.first + q * p.step] : make_int4(-3, -3, -3, -3);
      s_cut = (q < p.count && p.first + q * p.step < p.n_cut) ? 1 : 0;
      s_next = 0;
      s_ntouched = 0;
    }
    if (lane == 0) s_wtouched[wave] = 0;
    __syncthreads();
    const int4 tk = s_task;
    if (tk.x == -3) break;
    const bool use_cut = s_cut != 0;  // (the ends at or above p.cut of this task's centres belong to house_lds_kernel)
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
            // D = sum over the common neighbours w of v0, v1 of t(v1, w).  Round 6: no intersection -- summed over the graph a triangle a < b < c
            // is met as (v0, v1; w) = (b, a; c), (c, a; b), (c, b; a) and adds t(a, c) + 2 t(a, b), so an edge p < q collects its own t once per
            // common neighbour between p and q and twice per common neighbour above q, and both numbers are in the tables: below q =
            // tlt(q -> p), below p = tlt(p -> q).  (Until round 6 this was one bisected intersection per edge: 0.3 s of the house on R-MAT-20.)
            if (sb && te > 0u) {
              const unsigned tl = p.tlt[r0 + i];                                        // common neighbours below v0
              const unsigned tl_rev = p.tlt[r1 + lower_bound(col + r1, d1, v0)];     // ... below v1 (the entry v1 -> v0)
              res -= (unsigned long long)te * (2ull * (unsigned long long)(te - tl) + (unsigned long long)(tl - tl_rev));  // D
            }
          } else {
            L.cnt[lane] = (valid && v1 < v0) ? te : 0u;  // delta of this entry's 2-paths
            wave_sync();
            const int llen = !valid ? 0 : use_cut ? p.bnd0[(size_t)v1 * (size_t)p.bnd_stride] - r1 : d1;
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

// ---- house: the (count | weighted sum) maps of the heavy centres in LDS (round 6) --------------------------------------------------------
// house_acc_kernel's phase 1 is the rectangle's walk with a 64-bit word per end w -- n(w) 2-paths in the low 24 bits, S(w) = the sum of their
// rows' weights above -- and A(v0) = sum_w n(w) S(w) (the running form there: a 2-path of weight d adds d * n_old + S_old + d).  It walks
// EVERY 2-path of every centre (sum_x d(x)^2: 70 G on R-MAT-20 ef 16, 3.6 s at the rate of the L2 atomics), so the scheme of rect_lds_kernel
// pays even more: ranges of 16 K ids (one 64-bit LDS word each) from the last id down, (centre, range) tasks for the centres with more
// neighbours than a workgroup has threads, one task for all ranges of the others; what lies below the ranges stays with house_acc_kernel.
constexpr int kHouseLongQ = 192, kHouseLongWg = 1024;  // queue entries; keys from which a row is shared out
struct alignas(16) HouseLds {
  unsigned long long map[kHouseLdsIds];
  WaveLdsFlat w[kRectLdsWaves];
  unsigned delta[kRectLdsWaves][GM_WAVE];  // weight of the row a lane holds
  int4 lq[kHouseLongQ];                    // one-task centres: the rows with many keys in the current ranges {start, keys, weight}, walked by ALL waves
  int nlong;
  int2 task;
  int next;
  unsigned any[2];
};

__global__ __launch_bounds__(256) void house_bounds_kernel(GraphView g, HouseLdsRanges rr, int *__restrict__ bnd) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= g.nv) return;
  const int r = g.rp[x], d = g.rp[x + 1] - r;
  int at = 0;
  for (int k = 0; k <= rr.n; ++k) {
    at += (rr.rb(k) >= g.nv) ? d - at : lower_bound(g.col + r + at, d - at, rr.rb(k));
    bnd[(size_t)x * (size_t)(rr.n + 1) + (size_t)k] = r + at;
  }
}

// per centre: [0, nv) every 2-path (house_work_kernel), [nv, 2 nv) those whose end lies below the cut
__global__ __launch_bounds__(256) void house_work_cut_kernel(GraphView g, const int *__restrict__ bnd0, int bnd_stride, unsigned long long *__restrict__ work) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.nv) return;
  unsigned long long w = 0, wc = 0;
  for (int i = g.rp[v]; i < g.rp[v + 1]; ++i) {
    const int x = g.col[i];
    w += (unsigned long long)(g.rp[x + 1] - g.rp[x]) + 1ull;
    wc += (unsigned long long)(bnd0[(size_t)x * (size_t)bnd_stride] - g.rp[x]);
  }
  work[v] = w;
  work[(size_t)g.nv + (size_t)v] = wc;
}

// flat_walk with the owner of every key handed to f(in[], key[], own[] /* lane + 1 */)
template <int TG, class LT, class F>
__device__ __forceinline__ void flat_walk_own(LT &L, const int *__restrict__ col, const int lane, const int llen, const int kb, F f) {
  static_assert(kMarkWindow % (GM_WAVE * TG) == 0, "a window of owner marks is a whole number of tile groups");
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  if (total == 0) return;  // wave-uniform
  const int off = incl - llen;
  L.desc[lane] = make_int4(kb, off, 0, 0);
  unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
  int carry = 0;
  int key[TG], kown[TG];
  bool in[TG], have = false;
  for (int wb = 0; wb < total; wb += kMarkWindow) {
    const int wn = min(kMarkWindow, total - wb);
    const int nwords = ((wn + GM_WAVE * TG - 1) / (GM_WAVE * TG)) * (GM_WAVE * TG / 4);
    for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
    wave_sync();
    if (llen > 0 && off >= wb && off < wb + kMarkWindow) L.marks[off - wb] = (unsigned char)(lane + 1);
    wave_sync();
    for (int t = 0; t < wn; t += GM_WAVE * TG) {
      int own[TG], nkey[TG];
      bool nin[TG];
#pragma unroll
      for (int q = 0; q < TG; ++q) own[q] = (int)L.marks[t + q * GM_WAVE + lane];
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        own[q] = max(wave_incl_scan_max(own[q]), carry);
        carry = readlane(own[q], GM_WAVE - 1);
      }
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        const int pp = wb + t + q * GM_WAVE + lane;
        nin[q] = pp < total;
        const int4 d = L.desc[nin[q] ? own[q] - 1 : 0];
        nkey[q] = col[nin[q] ? d.x + (pp - d.y) : 0];
      }
      if (have) f(in, key, kown);
#pragma unroll
      for (int q = 0; q < TG; ++q) {
        key[q] = nkey[q];
        in[q] = nin[q];
        kown[q] = nin[q] ? own[q] : 1;
      }
      have = true;
    }
    wave_sync();
  }
  if (have) f(in, key, kown);
}

__global__ __launch_bounds__(kRectLdsWaves *GM_WAVE) void house_lds_kernel(const HouseLdsParams p) {
  __shared__ HouseLds S;
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  constexpr int nthreads = kRectLdsWaves * GM_WAVE;
  WaveLdsFlat &L = S.w[wave];
  unsigned *dl = S.delta[wave];
  for (int i = tid; i < kHouseLdsIds; i += nthreads) S.map[i] = 0ull;
  unsigned long long res = 0;  // modulo 2^64, like house_acc_kernel's
  const int bs = p.r.n + 1;
  auto fetch = [&]() {
    const unsigned long long q = atomicAdd(p.queue, 1ull);
    return (q < p.count) ? p.tasks[p.first + q * p.step] : make_int2(-3, -3);
  };
  if (tid == 0) {
    S.task = fetch();
    S.next = 0;
    S.nlong = 0;
    S.any[0] = S.any[1] = 0u;
  }
  __syncthreads();
  int ph = 0;
  // the keys (kb, llen) of this wave's 64 rows -- weights in dl[] -- into the map of range k; the centre itself is no end
  auto walk = [&](const int kb, int llen, const int k, const int v0, const bool RTN) {
    const int lo = p.r.rb(k);
    auto add = [&](const int key, const unsigned long long d) {
      if (key == v0) return;
      const unsigned long long inc = (d << 24) | 1ull;
      if (RTN) {
        const unsigned long long old = atomicAdd(&S.map[key - lo], inc);
        res += d * (old & 0xffffffull) + (old >> 24) + d;
      } else {
        atomicAdd(&S.map[key - lo], inc);
      }
    };
    unsigned long long lm = __ballot(llen >= kRectLdsLong);
    while (lm) {  // wave-uniform: a long row, the lanes stride it
      const int l = __ffsll((long long)lm) - 1;
      lm &= lm - 1ull;
      const int base = readlane(kb, l), len = readlane(llen, l);
      const unsigned long long d = (unsigned long long)dl[l];
      for (int j0 = 0; j0 < len; j0 += 4 * GM_WAVE) {
        int key[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) key[q] = col[base + min(j0 + q * GM_WAVE + lane, len - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j0 + q * GM_WAVE + lane < len) add(key[q], d);
      }
    }
    if (llen >= kRectLdsLong) llen = 0;
    auto inc = [&](const bool *in, const int *key, const int *own) {
#pragma unroll
      for (int q = 0; q < kRectLdsTiles; ++q)
        if (in[q]) add(key[q], (unsigned long long)dl[own[q] - 1]);
    };
    flat_walk_own<kRectLdsTiles>(L, col, lane, llen, kb, inc);
  };
  auto finish = [&](const unsigned any, const int k, const int2 *nt, const bool RTN) {
    if (__ballot(any != 0u) != 0ull && lane == 0) S.any[ph & 1] = 1u;
    __syncthreads();
    if (tid == nthreads - 1) {
      if (nt) S.task = *nt;
      S.next = 0;
      S.any[(ph + 1) & 1] = 0u;
    }
    if (S.any[ph & 1]) {  // (workgroup-uniform)
      const int ids = p.r.rb(k + 1) - p.r.rb(k);
      if (RTN) {
        for (int i = tid; i < ids; i += nthreads) S.map[i] = 0ull;
      } else {
        for (int i = tid; i < ids; i += nthreads) {
          const unsigned long long v = S.map[i];
          if (v) {
            S.map[i] = 0ull;
            res += (v & 0xffffffull) * (v >> 24);  // n(w) S(w)
          }
        }
      }
    }
    __syncthreads();
    ++ph;
  };
  for (;;) {
    const int2 t = S.task;
    if (t.x == -3) break;
    int2 nt = make_int2(-3, -3);
    if (tid == nthreads - 1) nt = fetch();
    const int v0 = t.x;
    const int r0 = rp[v0], nitems = rp[v0 + 1] - r0;
    if (t.y >= 0) {  // one range of a centre with many neighbours
      const int k = t.y;
      unsigned any = 0u;
      for (;;) {
        int bi = 0;
        if (lane == 0) bi = atomicAdd(&S.next, 1);
        bi = readfirst(bi);
        if (bi * GM_WAVE >= nitems) break;
        const int i = bi * GM_WAVE + lane;
        int llen = 0, kb = 0;
        unsigned d = 0u;
        if (i < nitems) {
          const int x = col[r0 + i];
          const int *__restrict__ bx = p.bnd + (size_t)x * (size_t)bs + (size_t)k;
          kb = bx[0];
          llen = bx[1] - kb;
          d = x < v0 ? p.t[r0 + i] : 0u;
        }
        wave_sync();  // (the previous batch's walk is done with dl[])
        dl[lane] = d;
        wave_sync();
        any |= (unsigned)llen;
        walk(kb, llen, k, v0, false);
      }
      finish(any, k, &nt, false);
    } else {  // every range of a centre with at most one neighbour per thread, from the last range down.  Such a centre has at most 1024
              // neighbours, so n(w) <= 1024 and S(w) <= 1024 * 1023 fit 11 + 21 bits: 32-bit counters, TWO ranges of ids per walk
      const int row = lane * kRectLdsWaves + wave;
      const bool valid = row < nitems;
      const int x = valid ? col[r0 + row] : 0;
      const int *__restrict__ bx = p.bnd + (size_t)x * (size_t)bs;
      dl[lane] = (valid && x < v0) ? p.t[r0 + row] : 0u;
      wave_sync();
      // (at most 32 neighbours: n <= 32 and S <= 32 * 31 fit 6 + 10 bits -- 16-bit counters, two to a word, FOUR ranges per walk: most
      // one-task centres are low-degree vertices beside a hub, and what a walk costs them is its ~2.5 us, not its keys)
      unsigned *map32 = reinterpret_cast<unsigned *>(S.map);
      const bool fw16 = nitems <= 32;
      const int rw = fw16 ? 4 : 2;                // ranges per walk
      const unsigned nb = fw16 ? 6u : 11u;        // bits of n in a counter
      const unsigned nm = (1u << nb) - 1u, fm = fw16 ? 0xffffu : 0xffffffffu;
      int k1 = p.r.n;  // ranges [k0, k1) per walk
      int ke = valid ? bx[k1] : 0;
      int kb = valid ? bx[max(k1 - rw, 0)] : 0;
      bool first = true;
      while (k1 > 0) {
        const int k0 = max(k1 - rw, 0);
        const int nkb = (valid && k0 > 0) ? bx[max(k0 - rw, 0)] : 0;
        int llen = ke - kb;
        const int lo = p.r.rb(k0);
        auto add = [&](const int key, const unsigned d) {
          if (key == v0) return;
          const unsigned off = (unsigned)(key - lo);
          const unsigned sh = fw16 ? (off & 1u) << 4 : 0u;
          const unsigned f = (atomicAdd(&map32[fw16 ? off >> 1 : off], ((d << nb) | 1u) << sh) >> sh) & fm;
          res += (unsigned long long)(d * (f & nm) + (f >> nb) + d);
        };
        const unsigned any = (unsigned)llen;
        // A centre beside a hub has the hub's row among its few: 10^5 keys that ONE wave would walk while fifteen wait.  Rows with
        // kHouseLongWg keys or more inside these ranges go to a queue and every wave takes every 16th slice of 256 keys of each.
        {
          const unsigned long long wm = __ballot(llen >= kHouseLongWg);
          if (wm != 0ull) {  // wave-uniform
            int slot0 = 0;
            if (lane == 0) slot0 = atomicAdd(&S.nlong, (int)__popcll(wm));
            slot0 = readfirst(slot0);
            if (llen >= kHouseLongWg) {
              const int slot = slot0 + rank_below(wm);
              if (slot < kHouseLongQ) {
                S.lq[slot] = make_int4(kb, llen, (int)dl[lane], 0);
                llen = 0;
              }
            }
          }
        }
        __syncthreads();
        const int nl = min(S.nlong, kHouseLongQ);
        for (int e = 0; e < nl; ++e) {
          const int4 le = S.lq[e];
          const unsigned d = (unsigned)le.z;
          for (int j0 = wave * 4 * GM_WAVE; j0 < le.y; j0 += kRectLdsWaves * 4 * GM_WAVE) {
            int key[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = col[le.x + min(j0 + q * GM_WAVE + lane, le.y - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j0 + q * GM_WAVE + lane < le.y) add(key[q], d);
          }
        }
        unsigned long long lm = __ballot(llen >= kRectLdsLong);
        while (lm) {  // wave-uniform: a long row, the lanes stride it
          const int l = __ffsll((long long)lm) - 1;
          lm &= lm - 1ull;
          const int base = readlane(kb, l), len = readlane(llen, l);
          const unsigned d = dl[l];
          for (int j0 = 0; j0 < len; j0 += 4 * GM_WAVE) {
            int key[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) key[q] = col[base + min(j0 + q * GM_WAVE + lane, len - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j0 + q * GM_WAVE + lane < len) add(key[q], d);
          }
        }
        if (llen >= kRectLdsLong) llen = 0;
        auto inc = [&](const bool *in, const int *key, const int *own) {
#pragma unroll
          for (int q = 0; q < kRectLdsTiles; ++q)
            if (in[q]) add(key[q], dl[own[q] - 1]);
        };
        flat_walk_own<kRectLdsTiles>(L, col, lane, llen, kb, inc);
        // (finish, for two ranges of 32-bit counters)
        if (__ballot(any != 0u) != 0ull && lane == 0) S.any[ph & 1] = 1u;
        __syncthreads();
        if (tid == nthreads - 1) {
          if (first) S.task = nt;
          S.next = 0;
          S.nlong = 0;  // (every wave has read the queue: the next walk appends after the barrier below)
          S.any[(ph + 1) & 1] = 0u;
        }
        if (S.any[ph & 1]) {
          uint4 *m4 = reinterpret_cast<uint4 *>(S.map);
          const int ids = p.r.rb(k1) - lo;
          const int words = fw16 ? (ids + 1) >> 1 : ids;
          for (int i = tid; i < ((words + 3) >> 2); i += nthreads) m4[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        ++ph;
        first = false;
        ke = kb;
        kb = nkb;
        k1 = k0;
      }
      wave_sync();
    }
  }
  const unsigned long long s0 = wave_sum_u64(res);
  if (lane == 0 && s0) atomicAdd(&p.counters[0], s0);
}

hipError_t launch_house_lds(const HouseLdsParams &p, int grid_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(house_lds_kernel, dim3((unsigned)grid_blocks), dim3(kRectLdsWaves * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}
hipError_t launch_house_bounds(const GraphView &g, const HouseLdsRanges &r, int *bnd, hipStream_t stream) {
  hipLaunchKernelGGL(house_bounds_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, r, bnd);
  return hipGetLastError();
}
hipError_t launch_house_work_cut(const GraphView &g, const int *bnd0, int bnd_stride, unsigned long long *work, hipStream_t stream) {
  hipLaunchKernelGGL(house_work_cut_kernel, dim3((unsigned)((g.nv + 255) / 256)), dim3(256), 0, stream, g, bnd0, bnd_stride, work);
  return hipGetLastError();
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
    case PAT_CLIQUEK:
      if (p.k > 8) hipLaunchKernelGGL(mine_kernel<PAT_CLIQUEK_DEEP>, grid, block, 0, stream, p);
      else hipLaunchKernelGGL(mine_kernel<PAT_CLIQUEK>, grid, block, 0, stream, p);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_mine_kernel() {}
void gm_touch_mine() { hipLaunchKernelGGL(gm_touch_mine_kernel, dim3(1), dim3(1), 0, 0); }

