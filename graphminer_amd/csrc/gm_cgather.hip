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
constexpr int kCgUnroll = 8;  // tiles of a row whose gathers are in flight together

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
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(w_out) : "s"((unsigned)m), "s"(2 * t) : "m0");
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(w_out) : "s"((unsigned)(m >> 32)), "s"(2 * t + 1) : "m0");
}

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
    for (int i = k0 + wave; i < d; i += kCgWaves) {
      const int rowo = readfirst(S.info[i].y) * p.core_words * 4;  // (scalar: < 2^32 bytes, launch_cgather checks)
      unsigned w_out = 0u;  // lane L: word L of the row (the words below the diagonal are zero)
      const int t0 = (i + 1) >> 6;
      if (t0 < ntiles) {  // the tile of the diagonal
        const int2 inf = S.info[t0 * GM_WAVE + lane];
        cg_tile<true>((unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf.x, rowo, 0), inf, t0, i, d, lane, w_out);
      }
      int tb = t0 + 1;
      for (; tb + kCgUnroll < ntiles; tb += kCgUnroll) {  // whole tiles beyond it, kCgUnroll gathers in flight
        int2 inf[kCgUnroll];
        unsigned w[kCgUnroll];
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) inf[k] = S.info[(tb + k) * GM_WAVE + lane];
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) w[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf[k].x, rowo, 0);
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) cg_tile<false>(w[k], inf[k], tb + k, i, d, lane, w_out);
      }
      for (; tb < ntiles; ++tb) {  // the rest, the row's last tile among them
        const int2 inf = S.info[tb * GM_WAVE + lane];
        cg_tile<true>((unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf.x, rowo, 0), inf, tb, i, d, lane, w_out);
      }
      if (lane < stride) mu[(size_t)i * stride + lane] = w_out;
    }
    __syncthreads();  // the column table is rewritten by the next vertex
  }
}

// ---- the core rows shared out to the XCDs (round 5) ------------------------------------------------------------------------------
// The gathers are bound by the lines they pull past L2 (53 GB per launch on the com-Orkut stand-in at an L2 hit rate of 0.26): the 128 MB
// core does not fit an XCD's 4 MB L2.  Shared out -- the workgroups that run on XCD k gather only the rows whose core row is = k mod 8 --
// an L2 sees an eighth of the core: round 4 measured 35 GB (hit rate 0.44) but lost the gain to a dequeue -> header -> column table ->
// barrier prologue per (matrix, share) with an eighth of a matrix's rows behind it.  Here a work unit is a BATCH of matrices x one share:
// the column tables of up to kCgBatchCols columns (2 .. 16 matrices, by degree class) sit in LDS together, the rows of the share are
// listed once (ballot + rank) and the waves take them from that list.  A workgroup whose own share has no batch left helps the next
// one, so the result does not depend on where workgroups land.
constexpr int kCgBatchCols = 4096;   // columns of a batch (sum of the matrices' padded widths)
constexpr int kCgBatchMax = 16;      // matrices of a batch
struct alignas(16) CGatherShareLds {
  int pj[kCgBatchCols + GM_WAVE];          // per column: s_j - core_base (< 0: below the core)
  unsigned short row_m[kCgBatchCols];      // the rows of this share: matrix of the batch ...
  unsigned short row_i[kCgBatchCols];      // ... and row of the matrix
  int ru[kCgBatchMax], d[kCgBatchMax], col0[kCgBatchMax], k0[kCgBatchMax];
  unsigned long long base[kCgBatchMax];
  int n_rows, next_row, unit, share;
};

__device__ __forceinline__ int cg_xcd_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7;
}

__global__ __launch_bounds__(kCgWaves *GM_WAVE) void cgather_share_kernel(const CGatherParams p) {
  __shared__ CGatherShareLds S;
  constexpr int NT = kCgWaves * GM_WAVE;
  const int tid = threadIdx.x, lane = tid & (GM_WAVE - 1), wave = readfirst(tid >> 6);
  const __amdgpu_buffer_rsrc_t core = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(p.core), 0, (int)p.core_bytes, 0x00020000);
  const int my = cg_xcd_id();
  int tried = 0, share = my;
  for (;;) {
    if (tid == 0) {
      int u_ = -1, k_ = share, t_ = tried;
      while (t_ < 8) {  // this XCD's share first, then the next ones
        const unsigned q = atomicAdd(p.queue + k_, 1u);
        if (q < (unsigned)p.n_units) { u_ = (int)q; break; }
        k_ = (k_ + 1) & 7;
        ++t_;
      }
      S.unit = u_;
      S.share = (k_ << 8) | t_;
    }
    __syncthreads();
    const int unit = S.unit;
    share = S.share >> 8;
    tried = S.share & 255;
    if (unit < 0) break;
    // unit -> segment (degree class) -> slots
    int seg = 0, ub = unit;
    while (seg < 2 && ub >= p.seg_units[seg]) { ub -= p.seg_units[seg]; ++seg; }
    const int bsz = p.seg_batch[seg];
    const int s0 = p.seg_first[seg] + ub * bsz, s1 = min(s0 + bsz, p.seg_first[seg] + p.seg_count[seg]);
    const int nm = s1 - s0;
    if (tid < nm) {
      const int slot = p.first_slot + s0 + tid;
      const int u = p.verts[slot];
      const int ru = p.rp[u], d = p.rp[u + 1] - ru;
      S.ru[tid] = ru;
      S.d[tid] = d;
      S.base[tid] = p.base[slot];
      S.k0[tid] = lower_bound(p.col + ru, d, p.core_base);
    }
    if (tid == 0) { S.n_rows = 0; S.next_row = 0; }
    __syncthreads();
    if (tid == 0) {
      int c = 0;
      for (int m = 0; m < nm; ++m) { S.col0[m] = c; c += (S.d[m] + GM_WAVE - 1) & ~(GM_WAVE - 1); }
    }
    __syncthreads();
    for (int m = 0; m < nm; ++m) {  // column tables + the rows of this share
      const int ru = S.ru[m], d = S.d[m], c0 = S.col0[m], k0 = S.k0[m];
      const int dpad = (d + GM_WAVE - 1) & ~(GM_WAVE - 1);
      for (int j = tid; j < dpad; j += NT) {
        const int pj = j < d ? p.col[ru + j] - p.core_base : -1;
        S.pj[c0 + j] = pj;
        const bool mine = j >= k0 && j < d && (pj & 7) == share;
        const unsigned long long mm = __ballot(mine);
        int b0 = 0;
        if (lane == 0 && mm) b0 = atomicAdd(&S.n_rows, (int)__popcll(mm));
        b0 = readfirst(b0);
        if (mine) {
          const int r = b0 + rank_below(mm);
          S.row_m[r] = (unsigned short)m;
          S.row_i[r] = (unsigned short)j;
        }
      }
    }
    __syncthreads();
    const int n_rows = S.n_rows;
    for (;;) {
      int r = 0;
      if (lane == 0) r = atomicAdd(&S.next_row, 1);
      r = readfirst(r);
      if (r >= n_rows) break;
      const int m = S.row_m[r], i = S.row_i[r];
      const int d = S.d[m], c0 = S.col0[m], stride = (d + 31) >> 5;
      const int ntiles = (d + GM_WAVE - 1) >> 6;
      const int rowo = readfirst(S.pj[c0 + i]) * p.core_words * 4;
      unsigned w_out = 0u;
      const int t0 = (i + 1) >> 6;
      auto info_of = [&](const int t) {
        const int pj = S.pj[c0 + t * GM_WAVE + lane];
        return make_int2(pj >= 0 ? (pj >> 5) << 2 : 0, pj);
      };
      if (t0 < ntiles) {
        const int2 inf = info_of(t0);
        cg_tile<true>((unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf.x, rowo, 0), inf, t0, i, d, lane, w_out);
      }
      int tb = t0 + 1;
      for (; tb + kCgUnroll < ntiles; tb += kCgUnroll) {
        int2 inf[kCgUnroll];
        unsigned w[kCgUnroll];
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) inf[k] = info_of(tb + k);
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) w[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf[k].x, rowo, 0);
#pragma unroll
        for (int k = 0; k < kCgUnroll; ++k) cg_tile<false>(w[k], inf[k], tb + k, i, d, lane, w_out);
      }
      for (; tb < ntiles; ++tb) {
        const int2 inf = info_of(tb);
        cg_tile<true>((unsigned)__builtin_amdgcn_raw_buffer_load_b32(core, inf.x, rowo, 0), inf, tb, i, d, lane, w_out);
      }
      if (lane < stride) (p.mat + S.base[m])[(size_t)i * stride + lane] = w_out;
    }
    __syncthreads();  // the tables are rewritten by the next unit
  }
}

hipError_t launch_cgather(const CGatherParams &p, int grid_blocks, hipStream_t stream) {
  static_assert(kCbMaxDeg <= 2048, "a row is at most 64 words: one per lane");
  if (p.core == nullptr || p.mat == nullptr || p.core_bytes == 0 || p.core_bytes > 0xffffffffull) return hipErrorInvalidValue;
  if (p.n_units > 0) hipLaunchKernelGGL(cgather_share_kernel, dim3((unsigned)grid_blocks), dim3(kCgWaves * GM_WAVE), 0, stream, p);
  else hipLaunchKernelGGL(cgather_kernel, dim3((unsigned)grid_blocks), dim3(kCgWaves * GM_WAVE), 0, stream, p);
  return hipGetLastError();
}
// workgroups per CU the launch asks for (GM_CG_PER_CU: sweeps).  The gathers are bound by the lines they pull through L2, yet they want
// every wave a CU has: 4 / 2 workgroups per CU 27.7 / 35.6 ms for the whole pattern.
int cgather_per_cu() {
  static const int v = [] {
    const char *e = gm_sweep_env("GM_CG_PER_CU");
    const int cap = (int)std::min<size_t>(163840 / std::max(sizeof(CGatherLds), sizeof(CGatherShareLds)), 2048 / (kCgWaves * GM_WAVE));
    return e ? std::max(1, std::min(atoi(e), cap)) : cap;
  }();
  return v;
}

}  // namespace gm

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_cgather_kernel() {}
void gm_touch_cgather() { hipLaunchKernelGGL(gm_touch_cgather_kernel, dim3(1), dim3(1), 0, 0); }
