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
//  * worker = one wave64. A worker dequeues TASK CHUNKS (a contiguous vertex range owning ~T CSR
//    entries), loads the chunk's row offsets and adjacency slice into LDS with coalesced loads
//    ("adjacency slices staged in LDS"), then walks the chunk 64 edges (u,v) at a time.
//  * the reference gives each edge a whole 32-lane warp, so on LiveJournal (mean oriented list
//    length 8.8) most lanes idle. Here the 64 edges of a batch are FLATTENED: the lookup lists of
//    all 64 edges form one virtual array, lane p handles element p. The element->edge map is an
//    owner-mark array in LDS resolved by a DPP max-scan, so every lane of every tile does one
//    binary search regardless of how skewed the list lengths are.
//  * per edge the cheaper direction is chosen: (X) stream N(v) with coalesced loads and bisect
//    the LDS-resident N(u), or (Y) take the keys from N(u) and bisect N(v) in global memory.
//  * counts stay in 64-bit per-lane registers; one global atomic per wave per counter at the end.
//  * 4-clique keeps the candidate sets as an LDS bit-matrix over N+(u) (row = edge (u,v1), bit j =
//    N+(u)[j] in N+(v1)); the second DFS level is popcount(row_i & row_j) without touching HBM.
#include "gm_mine.h"
#include "gm_setops.h"

namespace gm {

template <int PAT>
struct alignas(16) WaveLds {
  int4 desc[GM_WAVE];                     // per-edge descriptors of the current pass
  int stage[kStageCap];                   // staged adjacency slice col[e_begin .. e_end)
  int rpl[kMaxChunkVerts + 8];            // row offsets of the chunk's vertices (absolute)
  unsigned char marks[kMarkWindow];       // owner marks of the flattened positions
  unsigned cnt[GM_WAVE];                  // per-edge match counts (diamond)
  unsigned bits[PAT == PAT_CLIQUE4 ? kBitWords : 4];
};

struct Acc {
  unsigned long long c0 = 0, c1 = 0, c2 = 0;
};

__device__ __forceinline__ int bitlen(int x) { return 32 - __clz(x); }

// One flattened pass over the 64 edges of a batch.
//   llen       lookup-list length of this lane's edge (0 = edge not in this pass)
//   key_base   index into col[] of the lookup list
//   s_base     search list: index into L.stage (SLDS) or col[] (!SLDS)
//   s_len_flag search-list length | flag << 30
// act(found, owner_lane, key_index, pos_in_search_list, flag)
template <bool SLDS, int PAT, class Act>
__device__ __forceinline__ void flat_pass(WaveLds<PAT> &L, const int *__restrict__ col, const int lane, const int llen,
                                          const int key_base, const int s_base, const int s_len_flag, Act act) {
  const int incl = wave_incl_scan_add(llen);
  const int total = readlane(incl, GM_WAVE - 1);
  if (total == 0) return;  // wave-uniform
  const int off = incl - llen;
  L.desc[lane] = make_int4(key_base, off, s_base, s_len_flag);
  unsigned *m32 = reinterpret_cast<unsigned *>(L.marks);
  int carry = 0;
  for (int wb = 0; wb < total; wb += kMarkWindow) {
    const int wn = min(kMarkWindow, total - wb);
    const int nwords = ((wn + 63) >> 6) << 4;
    for (int i = lane; i < nwords; i += GM_WAVE) m32[i] = 0u;
    wave_sync();
    if (llen > 0 && off >= wb && off < wb + kMarkWindow) L.marks[off - wb] = (unsigned char)(lane + 1);
    wave_sync();
    for (int t = 0; t < wn; t += GM_WAVE) {
      int own = (int)L.marks[t + lane];
      own = max(wave_incl_scan_max(own), carry);
      carry = readlane(own, GM_WAVE - 1);
      const int p = wb + t + lane;
      if (p < total) {
        const int4 d = L.desc[own - 1];
        const int kidx = p - d.y;
        const int key = col[d.x + kidx];
        const int slen = d.w & 0x3fffffff;
        int pos;
        bool f;
        if (SLDS) f = contains(&L.stage[d.z], slen, key, &pos);
        else f = contains(col + d.z, slen, key, &pos);
        act(f, own - 1, kidx, pos, d.w >> 30);
      }
    }
    wave_sync();
  }
}

template <int PAT, bool BLDS>
__device__ __forceinline__ unsigned long long clique4_count(WaveLds<PAT> &L, const unsigned *__restrict__ bits, const int lane,
                                                            const int eb, const int nel, const int nvl, const int stride) {
  // sum_i sum_{j in M[i]} popc(M[i] & M[j])  ==  sum_{(v0,v1)} sum_{v2 in S1} |S1 ^ N+(v2)|
  unsigned long long c = 0;
  for (int le = lane; le < nel; le += GM_WAVE) {
    const int e = eb + le;
    int lo = 0, hi = nvl - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (L.rpl[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int row0 = L.rpl[lo] - eb;  // local index of edge (u, A[0])
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

template <int PAT>
__device__ __forceinline__ void process_chunk(const MineParams &p, WaveLds<PAT> &L, const ChunkRec r, const int lane,
                                              const int wave_slot, Acc &acc) {
  const int *__restrict__ rp = p.g.rp;
  const int *__restrict__ col = p.g.col;
  const int ub = r.u_begin, nvl = r.u_end - r.u_begin;
  const int eb = r.e_begin, nel = r.e_end - r.e_begin;

  for (int i = lane; i <= nvl; i += GM_WAVE) L.rpl[i] = rp[ub + i];
  wave_sync();
  const bool whole_rows = (eb == L.rpl[0]) && (r.e_end == L.rpl[nvl]);
  const bool staged = whole_rows && (nel <= kStageCap) && !(p.flags & 1);
  if (staged)
    for (int i = lane; i < nel; i += GM_WAVE) L.stage[i] = col[eb + i];

  // clique: adjacency bit-matrix of the chunk, one row of `stride` words per edge
  int stride = 0;
  bool bits_lds = true;
  unsigned *gbits = nullptr;
  if (PAT == PAT_CLIQUE4) {
    int m = 0;
    for (int i = lane; i < nvl; i += GM_WAVE) m = max(m, L.rpl[i + 1] - L.rpl[i]);
    stride = (wave_max_nonneg(m) + 31) >> 5;
    const long long words = (long long)nel * stride;
    bits_lds = words <= kBitWords;
    if (bits_lds) {
      for (int i = lane; i < (int)words; i += GM_WAVE) L.bits[i] = 0u;
    } else {
      gbits = p.scratch + (size_t)wave_slot * p.scratch_words;
      for (long long i = lane; i < words; i += GM_WAVE) gbits[i] = 0u;
    }
  }
  wave_sync();

  for (int le0 = 0; le0 < nel; le0 += GM_WAVE) {
    const int le = le0 + lane;
    const bool valid = le < nel;
    const int e = eb + le;
    int v = 0, u = 0, ru = 0, a = 0, rv = 0, b = 0, idx = 0;
    if (valid) {
      if (staged) v = L.stage[le];
      else v = col[e];
      int lo = 0, hi = nvl - 1;  // owner row: largest i with rpl[i] <= e
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (L.rpl[mid] <= e) lo = mid; else hi = mid - 1;
      }
      ru = L.rpl[lo];
      a = L.rpl[lo + 1] - ru;
      u = ub + lo;
      idx = e - ru;
      rv = rp[v];
      b = rp[v + 1] - rv;
    }
    // pattern-specific task filter / bounds
    bool act = valid;
    int al = a;       // effective length of A = N(u) (a prefix of the row)
    int flag = 0;
    if (PAT == PAT_DIAMOND) act = valid && (v < u);  // symmetry break, diamond.h:5
    if (PAT == PAT_MOTIF3) {
      al = idx;              // {w in N(v0) : w < v1} is exactly the idx entries before v1
      flag = (v < u) ? 1 : 0;
      if (valid) acc.c2 += (unsigned long long)idx;
    }
    act = act && al > 0 && b > 0;
    // direction: X streams B = N(v) and bisects A; Y takes keys from A and bisects B in HBM
    bool dirx = false;
    if (act) {
      if (staged) {
        const float cx = (float)b * (float)(2 + p.cost_x_step * bitlen(al));
        const float cy = (float)al * (float)(2 + p.cost_y_step * bitlen(b));
        dirx = cx <= cy;
      } else {
        dirx = b <= al;
      }
    }
    const bool diry = act && !dirx;
    if (PAT == PAT_DIAMOND) {
      L.cnt[lane] = 0u;
      wave_sync();
    }

    auto on_found = [&](bool f, int owner, int kidx, int pos, int fl, bool is_x) {
      if (!f) return;
      if (PAT == PAT_TC) {
        acc.c0 += 1;
      } else if (PAT == PAT_DIAMOND) {
        atomicAdd(&L.cnt[owner], 1u);
      } else if (PAT == PAT_MOTIF3) {
        acc.c0 += 1;               // |A' ^ B| summed over all directed edges
        acc.c1 += (unsigned)fl;    // ... over edges with v1 < v0  (triangles)
      } else if (PAT == PAT_CLIQUE4) {
        const int cbit = is_x ? pos : kidx;  // position of the common neighbour inside N+(u)
        const size_t word = (size_t)(le0 + owner) * stride + (cbit >> 5);
        if (bits_lds) atomicOr(&L.bits[word], 1u << (cbit & 31));
        else atomicOr(&gbits[word], 1u << (cbit & 31));
      }
    };

    // pass X
    {
      const int llen = dirx ? b : 0;
      const int s_len_flag = al | (flag << 30);
      auto actx = [&](bool f, int owner, int kidx, int pos, int fl) { on_found(f, owner, kidx, pos, fl, true); };
      if (staged) flat_pass<true, PAT>(L, col, lane, llen, rv, ru - eb, s_len_flag, actx);
      else flat_pass<false, PAT>(L, col, lane, llen, rv, ru, s_len_flag, actx);
    }
    // pass Y
    {
      const int llen = diry ? al : 0;
      const int s_len_flag = b | (flag << 30);
      auto acty = [&](bool f, int owner, int kidx, int pos, int fl) { on_found(f, owner, kidx, pos, fl, false); };
      flat_pass<false, PAT>(L, col, lane, llen, ru, rv, s_len_flag, acty);
    }

    if (PAT == PAT_DIAMOND) {
      wave_sync();
      const unsigned long long n = L.cnt[lane];
      acc.c0 += n * (n - 1) / 2;  // C(n,2), 64-bit (diamond_count.cuh:15-17)
      wave_sync();
    }
  }

  if (PAT == PAT_CLIQUE4) {
    wave_sync();
    if (!bits_lds) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");  // own atomics -> own loads through L2
    if (bits_lds) acc.c0 += clique4_count<PAT, true>(L, L.bits, lane, eb, nel, nvl, stride);
    else acc.c0 += clique4_count<PAT, false>(L, gbits, lane, eb, nel, nvl, stride);
    wave_sync();
  }
}

template <int PAT>
__global__ __launch_bounds__(kWavesPerBlock *GM_WAVE) void mine_kernel(const MineParams p) {
  __shared__ WaveLds<PAT> lds[kWavesPerBlock];
  const int lane = threadIdx.x & (GM_WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int wave_slot = blockIdx.x * kWavesPerBlock + wave;
  WaveLds<PAT> &L = lds[wave];
  Acc acc;
  for (;;) {
    unsigned q = 0;
    if (lane == 0) q = atomicAdd(p.queue, (unsigned)p.grab);
    q = (unsigned)readfirst((int)q);
    if (q >= (unsigned)p.count) break;
    const unsigned qe = min(q + (unsigned)p.grab, (unsigned)p.count);
    for (unsigned i = q; i < qe; ++i) {
      const ChunkRec r = p.chunks[(size_t)p.first + (size_t)i * (size_t)p.step];
      process_chunk<PAT>(p, L, r, lane, wave_slot, acc);
    }
  }
  const unsigned long long s0 = wave_sum_u64(acc.c0);
  const unsigned long long s1 = wave_sum_u64(acc.c1);
  const unsigned long long s2 = wave_sum_u64(acc.c2);
  if (lane == 0) {
    if (s0) atomicAdd(&p.counters[0], s0);
    if (s1) atomicAdd(&p.counters[1], s1);
    if (s2) atomicAdd(&p.counters[2], s2);
  }
}

size_t mine_lds_bytes(Pattern pat) {
  switch (pat) {
    case PAT_CLIQUE4: return sizeof(WaveLds<PAT_CLIQUE4>) * kWavesPerBlock;
    default: return sizeof(WaveLds<PAT_TC>) * kWavesPerBlock;
  }
}

hipError_t launch_mine(Pattern pat, const MineParams &p, int grid_blocks, hipStream_t stream) {
  dim3 grid((unsigned)grid_blocks), block(kWavesPerBlock * GM_WAVE);
  switch (pat) {
    case PAT_TC: hipLaunchKernelGGL(mine_kernel<PAT_TC>, grid, block, 0, stream, p); break;
    case PAT_DIAMOND: hipLaunchKernelGGL(mine_kernel<PAT_DIAMOND>, grid, block, 0, stream, p); break;
    case PAT_MOTIF3: hipLaunchKernelGGL(mine_kernel<PAT_MOTIF3>, grid, block, 0, stream, p); break;
    case PAT_CLIQUE4: hipLaunchKernelGGL(mine_kernel<PAT_CLIQUE4>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gm
