// gm_tables.hip -- task tables, built on the device: chunk tables (replaces Graph::init_edgelist, src/common/graph.cc:297-326, and the
// per-GPU COO copies of Scheduler::round_robin, src/common/scheduler.cc:34-85), edge descriptors, the task lists of gm_tch.hip, the
// two-phase plan of the wide k-clique vertices, and the multi-GPU split as index arithmetic (gm_partition).
#include "gm_host.h"
#include "gm_scan.h"
using namespace gm;

static void free_table(ChunkTable &t) {
  if (t.d) dev_free(t.d);
  if (t.d_slot) dev_free(t.d_slot);
  if (t.d_row_slot && t.own_bitmaps) dev_free(t.d_row_slot);
  for (int i = 0; i < 2; ++i) if (t.d_order[i]) dev_free(t.d_order[i]);
  if (t.d_bitmaps && t.own_bitmaps) dev_free(t.d_bitmaps);
  if (t.d_edges) dev_free(t.d_edges);
  if (t.d_firstv) dev_free(t.d_firstv);
  if (t.d_cost) dev_free(t.d_cost);
  for (auto &sh : t.shares)
    if (sh.d) dev_free(sh.d);
  t.shares.clear();
  t.d_edges = t.d_firstv = nullptr; t.d_cost = nullptr;
  t.d = nullptr; t.d_slot = nullptr; t.d_row_slot = nullptr; t.d_order[0] = t.d_order[1] = nullptr; t.d_bitmaps = nullptr;
}
void free_tables(gm_graph *g) {
  for (auto &t : g->tables) free_table(t);
  g->tables.clear();
}

// ------------------------------------------------------------------------------------------------
// task chunk tables
// ------------------------------------------------------------------------------------------------
// The greedy chunk walk over the vertices [v0, v1): a contiguous run of whole rows is closed when it reaches `target` entries,
// would exceed the LDS stage (or, k-clique, the bit-matrix budget), spans kMaxChunkVerts rows, or meets a row that is left out /
// too long for the stage; rows longer than the stage are SPLIT into `target`-entry chunks (allow_split) or chunks of their
// own. Runs on the host (gm_chunk_table, GM_HOST_TABLES) and, one thread per block of kTableBlock vertices, on the device:
// the walk restarts at every block boundary, so a table is the same whichever side built it.
constexpr int kTableBlock = 512;  // (2048 until round 3: four times fewer walkers, each four times as long -- the walk is a serial chain)
struct ChunkWalk {
  int target, allow_split, bit_words, stage_cap;
  RowFilter rf;
};
// (One loop over the vertices with the open chunk as STATE -- no nested loops: the device runs 32 walkers per wave, and with a loop per
// chunk inside the loop over vertices every walker sat in a branch of its own: 0.6 ms per pass over 4 M vertices, all of it divergence.)
template <class Emit>
__host__ __device__ inline unsigned long long walk_chunks(const ChunkWalk &w, const int *rp, int v0, int v1, Emit emit) {
  unsigned long long max_bit_words = 0;
  bool open = false;  // a chunk [start, u) of whole rows with `edges` > 0 entries is being filled
  int start = 0, edges = 0, maxd = 0;
  auto close = [&](int end) {
    emit(ChunkRec{start, end, rp[start], rp[end], 0, 1, GM_WAVE, 0});
    if (w.bit_words) {
      const unsigned long long bw = (unsigned long long)edges * (unsigned long long)((maxd + 31) / 32);
      if (bw > (unsigned long long)w.bit_words) max_bit_words = max(max_bit_words, bw);
    }
    open = false;
  };
  for (int u = v0; u < v1; ++u) {
    const int d = rp[u + 1] - rp[u];
    const bool left_out = d > 0 && w.rf.skips(d);
    if (open) {  // does row u still go into the open chunk?  (an empty row does, a row that is left out or too long ends it)
      bool joins = (u - start) < kMaxChunkVerts && d <= w.stage_cap && !left_out && edges + d <= w.stage_cap;
      if (joins && w.bit_words) joins = (long long)(edges + d) * ((max(maxd, d) + 31) / 32) <= (long long)w.bit_words;
      if (!joins) close(u);
    }
    if (!open) {  // row u starts a chunk, is a chunk (or several) of its own, or is passed over
      if (d == 0 || left_out) continue;
      if (d > w.stage_cap) {
        if (w.allow_split) {
          for (int s0 = rp[u]; s0 < rp[u + 1]; s0 += w.target) emit(ChunkRec{u, u + 1, s0, min(s0 + w.target, rp[u + 1]), 0, 1, GM_WAVE, 0});
        } else {
          emit(ChunkRec{u, u + 1, rp[u], rp[u + 1], 0, 1, GM_WAVE, 0});
          if (w.bit_words) max_bit_words = max(max_bit_words, (unsigned long long)d * (unsigned long long)((d + 31) / 32));
        }
        continue;
      }
      open = true;
      start = u;
      edges = 0;
      maxd = 0;
    }
    edges += d;
    maxd = max(maxd, d);
    if (edges >= w.target) close(u + 1);
  }
  if (open) close(v1);
  return max_bit_words;
}

static void build_chunks(const std::vector<int> &rp, int nv, int target, bool allow_split, int bit_words, int stage_cap,
                         std::vector<ChunkRec> &out, unsigned long long &max_bit_words, const RowFilter &rf = RowFilter()) {
  out.clear();
  max_bit_words = 0;
  ChunkWalk w{target, allow_split ? 1 : 0, bit_words, stage_cap, rf};
  for (int v0 = 0; v0 < nv; v0 += kTableBlock)
    max_bit_words = std::max(max_bit_words, walk_chunks(w, rp.data(), v0, std::min(v0 + kTableBlock, nv), [&](const ChunkRec &r) { out.push_back(r); }));
}

// one workgroup per hub row: set bit x for every neighbour x of the row
__global__ __launch_bounds__(256) void bitmap_build_kernel(const int *__restrict__ rp, const int *__restrict__ col,
                                                           const int *__restrict__ rows, unsigned *__restrict__ bitmaps,
                                                           unsigned long long words) {
  const int u = rows[blockIdx.x];
  unsigned *bm = bitmaps + (size_t)blockIdx.x * words;
  for (int i = rp[u] + threadIdx.x; i < rp[u + 1]; i += blockDim.x) {
    const unsigned x = (unsigned)col[i];
    atomicOr(&bm[x >> 5], 1u << (x & 31u));
  }
}

// one workgroup per chunk: estimated work = keys touched. DAG patterns: sum over the task edges (u, v) of d(u) + d(v);
// symmetric-graph patterns (owner_rule): only the edges whose longer row is u are tasks here, and each streams the
// shorter list, d(v) keys (process_chunk's ownership rule).
__global__ __launch_bounds__(256) void chunk_cost_kernel(const int *__restrict__ rp, const int *__restrict__ col,
                                                         const ChunkRec *__restrict__ chunks, unsigned long long *__restrict__ cost,
                                                         int owner_rule, int stage_cap, const int *__restrict__ trp = nullptr,
                                                         const int *__restrict__ tlen = nullptr, int tstride = 2, const int *__restrict__ kst_rp = nullptr) {
  const ChunkRec r = chunks[blockIdx.x];
  unsigned long long c = 0;
  if (owner_rule == 2) {  // gm_tch.hip: the keys of the lists this chunk's vertices host
    // (tlen: the length field of the first task record, tstride: ints per record -- int2 {start, len} of gm_tch.hip, CBuildTask of gm_cbuild.hip)
    // (gridDim.y workgroups share a chunk: a hub of R-MAT-22 hosts 2 * 10^5 tasks -- one workgroup walking them alone was 1.6 of the kernel's 1.7 ms)
    const int t0 = trp[r.u_begin], t1 = trp[r.u_end];
    const int ny = max(1, min((int)gridDim.y, (t1 - t0 + 1023) >> 10));  // workgroups that take part: one per 1024 tasks
    if ((int)blockIdx.y >= ny) return;
    // (kst_rp: trp / tlen hold the longer lists only; the keys of the short ones are the hosts' range of the key stream -- no per-task cost there)
    if (kst_rp && blockIdx.y == 0 && threadIdx.x == 0) c += (unsigned long long)(kst_rp[r.u_end] - kst_rp[r.u_begin]);
    for (int te = t0 + (int)blockIdx.y * 256 + (int)threadIdx.x; te < t1; te += 256 * ny) c += (unsigned long long)tlen[(size_t)te * (size_t)tstride] + 8ull;
  } else if (owner_rule) {
    // a key streamed by a SPLIT chunk is a random probe of the hub row's bitmap in HBM, a key of a staged chunk an LDS filter probe
    const unsigned long long w = (r.u_end == r.u_begin + 1 && (r.e_begin != rp[r.u_begin] || r.e_end != rp[r.u_end])) ? (unsigned long long)kProbeCost : 1ull;
    for (int u = r.u_begin; u < r.u_end; ++u) {  // (a SPLIT chunk has one row; a staged chunk few long or many short ones)
      const int a = rp[u + 1] - rp[u];
      const int lo = max(rp[u], r.e_begin), hi = min(rp[u + 1], r.e_end);
      for (int e = lo + (int)threadIdx.x; e < hi; e += 256) {
        const int v = col[e];
        const int b = rp[v + 1] - rp[v];
        if (sym_hosts(a, b, u, v, stage_cap)) c += ((unsigned long long)b + 8ull) * w;  // (X streams N(v) whichever is longer)
      }
    }
  } else {
    for (int e = r.e_begin + (int)threadIdx.x; e < r.e_end; e += 256) {
      const int v = col[e];
      c += (unsigned long long)(rp[v + 1] - rp[v]);
    }
    for (int u = r.u_begin + (int)threadIdx.x; u < r.u_end; u += 256) {
      const int lo = max(rp[u], r.e_begin), hi = min(rp[u + 1], r.e_end);
      c += (unsigned long long)max(hi - lo, 0) * (unsigned long long)(rp[u + 1] - rp[u]);
    }
  }
  c = gm::wave_sum_u64(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&cost[blockIdx.x], c);
}

// edges per batch of a chunk: 64, or kSplitBatch in the SPLIT chunks of the symmetric-graph patterns whose task edges stream
// long lists (estimated keys per entry >= kSplitBatchMinKeys)
static int batch_edges(const ChunkRec &r, const std::vector<int> &rp, int stage_cap, unsigned long long cost) {
  const bool whole = r.e_begin == rp[(size_t)r.u_begin] && r.e_end == rp[(size_t)r.u_end];
  const unsigned long long nel = (unsigned long long)std::max(r.e_end - r.e_begin, 1);
  // (the one-row chunks of the big-LDS classes are in the same situation as SPLIT chunks: every task edge streams a long list)
  const bool long_lists = cost / nel >= (unsigned long long)kSplitBatchMinKeys;
  return ((stage_cap == kStageCapWide && !whole && long_lists) || (stage_cap > kStageCapWide && long_lists)) ? kSplitBatch : GM_WAVE;
}

// ------------------------------------------------------------------------------------------------
// Task-chunk table, built on the device. The reference builds its COO task list with a serial host loop
// (Graph::init_edgelist, src/common/graph.cc:297-326) and round 1 of this library walked the vertices on the host too
// (~100 ms per table at nv = 2^24, three tables for a symmetric-graph pattern). Here:
//   greedy walk, one thread per block of kTableBlock vertices (count -> exclusive scan -> emit)  ->  cost kernel  ->
//   parts (scan + expand)  ->  dequeue orders (stable radix sorts)  ->  hub-row bitmaps (select + scatter)
// What stays on the host is O(chunks) or O(hub rows): the per-chunk edge prefix, the dequeue orders' host copies.
// (A fully parallel chunk definition -- short rows grouped by floor(prefix / G) -- was tried first: its groups cannot fill
// the stage as tightly as the greedy walk, the chunks of a flat graph shrank from ~1024 to ~683 entries and TC on the
// LiveJournal-size flat graph went from 0.873 to 1.145 ms; profiles/r02/ab_setup_device_tables.log.)
// ------------------------------------------------------------------------------------------------
// pass 0: chunks per vertex block (+ the k-clique arena requirement); pass 1: the records, at the block's offset.
// A workgroup owns kWalkPerWG blocks: all 256 threads copy the blocks' offsets into LDS with coalesced loads (64 KB of
// the CU's 160 KB), then lanes 0 .. kWalkPerWG-1 each walk one block out of LDS -- the walk is a serial chain of dependent
// reads, ~30 cycles per vertex from LDS against a memory round trip per cache line from HBM (one thread per block straight
// from memory: 132 ms for the three 3-motif tables of R-MAT-24; this form: see profiles/r02/ab_setup_device_tables.log).
constexpr int kWalkPerWG = 32;
__global__ __launch_bounds__(256) void tab_walk_kernel(ChunkWalk w, int nv, const int *__restrict__ rp, int nblocks, int *__restrict__ count,
                                                      unsigned long long *__restrict__ max_bw, const int *__restrict__ offset, ChunkRec *__restrict__ recs) {
  __shared__ int rpl[kWalkPerWG][kTableBlock + 1];  // (row stride 2049 words: the walkers' lanes fall into different banks)
  const int b0 = blockIdx.x * kWalkPerWG;
  for (int k = 0; k < kWalkPerWG; ++k) {
    const int v0 = (b0 + k) * kTableBlock;
    if (v0 >= nv) break;
    const int n = min(kTableBlock, nv - v0) + 1;
    for (int i = threadIdx.x; i < n; i += 256) rpl[k][i] = rp[v0 + i];  // (all four waves fill: a single wave's serial round trips made this 3-5 ms per pass)
  }
  __syncthreads();
  const int k = threadIdx.x, b = b0 + k;
  if (k >= kWalkPerWG || b > nblocks) return;
  if (b == nblocks) { if (!recs) count[b] = 0; return; }
  const int v0 = b * kTableBlock, v1 = min(v0 + kTableBlock, nv);
  const int *lrp = &rpl[k][0] - v0;  // indexed by the absolute vertex id
  if (!recs) {
    int n = 0;
    const unsigned long long bw = walk_chunks(w, lrp, v0, v1, [&](const ChunkRec &) { ++n; });
    count[b] = n;
    if (bw) atomicMax(max_bw, bw);
  } else {
    int o = offset[b];
    walk_chunks(w, lrp, v0, v1, [&](const ChunkRec &r) { recs[o++] = r; });
  }
}
struct TableDevParams {
  int nv;
  RowFilter rf;
};
__device__ __forceinline__ bool rf_skips(const RowFilter &rf, int d) { return rf.skips(d); }
// parts and batch sizes per chunk (batch_edges + the part rule of the host path)
__global__ __launch_bounds__(256) void tab_parts_kernel(int n0, const ChunkRec *__restrict__ recs, const int *__restrict__ rp,
                                                        const unsigned long long *__restrict__ cost, int stage_cap, unsigned long long cap,
                                                        int cut, int *__restrict__ np_out, int *__restrict__ bsz_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n0) return;
  if (c == n0) { np_out[c] = 0; return; }
  const ChunkRec r = recs[c];
  const bool whole = r.e_begin == rp[r.u_begin] && r.e_end == rp[r.u_end];
  const unsigned long long nel = (unsigned long long)max(r.e_end - r.e_begin, 1);
  const bool long_lists = cost[c] / nel >= (unsigned long long)kSplitBatchMinKeys;
  const int bsz = ((stage_cap == kStageCapWide && !whole && long_lists) || (stage_cap > kStageCapWide && long_lists)) ? kSplitBatch : GM_WAVE;
  int np = 1;
  if (cut) {
    const int batches = (r.e_end - r.e_begin + bsz - 1) / bsz;
    const int min_batches = stage_cap == kStageCapBig ? 64 : (stage_cap == kStageCapMid ? 32 : 1);
    const unsigned long long want = cost[c] / cap + (cost[c] % cap != 0);  // (no cost + cap - 1: it wraps for huge caps)
    np = (int)max(1ull, min((unsigned long long)max(batches / min_batches, 1), want));
  }
  np_out[c] = np;
  bsz_out[c] = bsz;
}
__global__ __launch_bounds__(256) void tab_expand_kernel(int n0, const ChunkRec *__restrict__ recs, const unsigned long long *__restrict__ cost,
                                                         const int *__restrict__ np_in, const int *__restrict__ bsz_in, const int *__restrict__ off,
                                                         ChunkRec *__restrict__ out, unsigned long long *__restrict__ cost_out,
                                                         int *__restrict__ edges_out, int *__restrict__ first_vertex) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n0) return;
  ChunkRec r = recs[c];
  const int np = np_in[c], bsz = bsz_in[c], nel = r.e_end - r.e_begin;
  r.nparts = np;
  r.batch = bsz;
  for (int qd = 0; qd < np; ++qd) {
    r.part = qd;
    int mine = 0;  // task edges of a part = the entries of its batches
    for (int b = qd; b * bsz < nel; b += np) mine += min(bsz, nel - b * bsz);
    const int o = off[c] + qd;
    out[o] = r;
    cost_out[o] = cost[c] / (unsigned long long)np;
    edges_out[o] = mine;
    first_vertex[o] = r.u_begin;
  }
}
__global__ __launch_bounds__(256) void sum_u64_kernel(long long n, const unsigned long long *__restrict__ x, unsigned long long *__restrict__ out) {
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += x[i];
  s = gm::wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
__global__ __launch_bounds__(256) void tab_totals_kernel(int n, const int *__restrict__ edges, const unsigned long long *__restrict__ cost,
                                                         unsigned long long *__restrict__ out) {
  unsigned long long se = 0, sc = 0, nz = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
    se += (unsigned long long)edges[c];
    sc += cost[c];
    nz += cost[c] != 0ull ? 1ull : 0ull;
  }
  se = gm::wave_sum_u64(se);
  sc = gm::wave_sum_u64(sc);
  nz = gm::wave_sum_u64(nz);
  if ((threadIdx.x & 63) == 0) {
    if (se) atomicAdd(&out[0], se);
    if (sc) atomicAdd(&out[1], sc);
    if (nz) atomicAdd(&out[2], nz);  // chunks with any work: the head of the cost-ordered dequeue list (d_order[1])
  }
}
// rev_rest: the chunks below the heavy mark go in DESCENDING chunk-id order. On a topologically numbered DAG ids ascend in degree, so
// ascending vertex order would end the launch on its heaviest ordinary chunks (TC on the power-law LiveJournal stand-in: 1.14 ms as
// numbered, degrees descending, vs 1.88 ms on the renumbered copy before this)
__global__ __launch_bounds__(256) void tab_orderkeys_kernel(int n, const unsigned long long *__restrict__ cost, unsigned long long heavy, int classes_only,
                                                            int rev_rest, unsigned long long *__restrict__ key0, unsigned long long *__restrict__ key1,
                                                            int *__restrict__ iota) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const unsigned long long x = cost[c];
  key0[c] = x >= heavy ? (classes_only ? 1ull : x) + (rev_rest ? (1ull << 56) : 0ull) : (rev_rest ? (unsigned long long)c : 0ull);
  key1[c] = x;
  iota[c] = c;
}
__global__ __launch_bounds__(256) void tab_hubflag_kernel(TableDevParams q, int bitmap_min_deg, const int *__restrict__ rp, int *__restrict__ flag, int *__restrict__ iota) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= q.nv) return;
  const int d = rp[v + 1] - rp[v];
  flag[v] = (d > bitmap_min_deg && !rf_skips(q.rf, d)) ? 1 : 0;
  iota[v] = v;
}
__global__ __launch_bounds__(256) void tab_rowslot_kernel(int nb, const int *__restrict__ rows, int *__restrict__ row_slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) row_slot[rows[i]] = i;
}
__global__ __launch_bounds__(256) void tab_chunkslot_kernel(int n, const ChunkRec *__restrict__ recs, const int *__restrict__ rp, int stage_cap,
                                                            const int *__restrict__ row_slot, int *__restrict__ slots) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const ChunkRec r = recs[c];
  slots[c] = (r.u_end == r.u_begin + 1 && rp[r.u_begin + 1] - rp[r.u_begin] > stage_cap) ? row_slot[r.u_begin] : -1;
}
__global__ __launch_bounds__(256) void gather_deg_kernel(int m, const int *__restrict__ verts, const int *__restrict__ rp, int *__restrict__ deg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) deg[i] = rp[verts[i] + 1] - rp[verts[i]];
}

// (trp / tlen / tstride: the task lists the costs of an rf.tct table are counted from; default: the graph's, gm_tch.hip)
static int build_table_device(gm_graph *g, ChunkTable &t, bool sym_table, double &bitmap_ms, const int *trp = nullptr, const int *tlen = nullptr,
                              int tstride = 2) {
  const int *kst_rp = nullptr;
  // (a stream without the rows of the hub core, gm_ctc.hip, prices the triangle count's own tables -- rf.tct == 2 -- only: the edge supports
  // walk every task)
  const bool kst_costs = g->d_kst_rp && (g->kst_skip_from >= g->nv ? t.rf.tct != 2 : (t.rf.tct == 2 || !g->d_trp));
  if (!trp && kst_costs) {  // the costs of the triangle count's tables: the key stream + the longer lists
    trp = g->d_trpl;
    tlen = &g->d_tdescl[0].y;
    tstride = 2;
    kst_rp = g->d_kst_rp;
  } else if (!trp) {
    trp = g->d_trp;
    tlen = g->d_tdesc ? &g->d_tdesc[0].y : nullptr;
    tstride = 2;
  }
  const int nv = g->nv;
  TableDevParams q;
  q.nv = nv;
  q.rf = t.rf;
  ScanTemp tmp;
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  // the greedy walk, one thread per block of kTableBlock vertices: count, scan, emit
  ChunkWalk w{t.target, t.allow_split ? 1 : 0, t.bit_words, t.stage_cap, t.rf};
  setup_trace("table: begin");
  const int nblk = (nv + kTableBlock - 1) / kTableBlock;
  DevBuf<int> bcount, boff;
  DevBuf<unsigned long long> maxbw;
  HIP_TRY(bcount.alloc((size_t)nblk + 1));
  HIP_TRY(boff.alloc((size_t)nblk + 1));
  HIP_TRY(maxbw.alloc(1));
  HIP_TRY(hipMemsetAsync(maxbw.p, 0, 8, 0));
  const dim3 wgrid((unsigned)((nblk + 1 + kWalkPerWG - 1) / kWalkPerWG));
  hipLaunchKernelGGL(tab_walk_kernel, wgrid, dim3(256), 0, 0, w, nv, g->d_rp, nblk, bcount.p, maxbw.p, (const int *)nullptr, (ChunkRec *)nullptr);
  HIP_TRY(dev_exclusive_sum(tmp, bcount.p, boff.p, (size_t)nblk + 1));
  int n0 = 0;
  HIP_TRY(hipMemcpy(&n0, boff.p + nblk, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&t.max_bit_words, maxbw.p, 8, hipMemcpyDeviceToHost));
  setup_trace("table: walk count + scan");
  t.n = 0;
  HIP_TRY(dev_malloc(&t.d, sizeof(ChunkRec)));  // (placeholder, replaced below when the table has chunks)
  if (n0 == 0) {
    t.edge_prefix.assign(1, 0ull);
    t.host_ready = true;
    return GM_OK;
  }
  DevBuf<ChunkRec> recs0;
  HIP_TRY(recs0.alloc((size_t)n0));
  hipLaunchKernelGGL(tab_walk_kernel, wgrid, dim3(256), 0, 0, w, nv, g->d_rp, nblk, bcount.p, maxbw.p, (const int *)boff.p, recs0.p);
  // estimated work per chunk, parts
  DevBuf<unsigned long long> cost0;
  HIP_TRY(cost0.alloc((size_t)n0));
  HIP_TRY(hipMemsetAsync(cost0.p, 0, sizeof(unsigned long long) * (size_t)n0, 0));
  hipLaunchKernelGGL(chunk_cost_kernel, dim3((unsigned)n0, t.rf.tct ? 16u : 1u), dim3(256), 0, 0, g->d_rp, g->d_col, recs0.p, cost0.p, t.rf.tct ? 2 : (sym_table ? 1 : 0), kStageCapWide,
                     trp, tlen, tstride, kst_rp);
  DevBuf<int> np, bsz, off;
  HIP_TRY(np.alloc((size_t)n0 + 1));
  HIP_TRY(bsz.alloc((size_t)n0 + 1));
  HIP_TRY(off.alloc((size_t)n0 + 1));
  hipLaunchKernelGGL(tab_parts_kernel, blocks(n0 + 1), dim3(256), 0, 0, n0, recs0.p, g->d_rp, cost0.p, t.stage_cap,
                     std::max<unsigned long long>(t.part_cap, 1), ((t.allow_split || sym_table) && t.part_cap != 0) ? 1 : 0, np.p, bsz.p);  // part_cap 0: never cut
  HIP_TRY(dev_exclusive_sum(tmp, np.p, off.p, (size_t)n0 + 1));
  int n = 0;
  HIP_TRY(hipMemcpy(&n, off.p + n0, sizeof(int), hipMemcpyDeviceToHost));
  setup_trace("table: walk emit, cost, parts");
  dev_free(t.d);
  t.d = nullptr;
  HIP_TRY(dev_malloc(&t.d, sizeof(ChunkRec) * (size_t)n));
  setup_trace("table: free + alloc of the records");
  // per-chunk scalars (cost, task edges, first vertex): they stay on the device, the host takes the two totals (table_host_views: the rest, on demand)
  HIP_TRY(dev_malloc(&t.d_cost, sizeof(unsigned long long) * (size_t)n));
  HIP_TRY(dev_malloc(&t.d_edges, sizeof(int) * (size_t)n));
  HIP_TRY(dev_malloc(&t.d_firstv, sizeof(int) * (size_t)n));
  DevBuf<unsigned long long> totals;
  HIP_TRY(totals.alloc(3));
  HIP_TRY(hipMemsetAsync(totals.p, 0, 24, 0));
  hipLaunchKernelGGL(tab_expand_kernel, blocks(n0), dim3(256), 0, 0, n0, recs0.p, cost0.p, np.p, bsz.p, off.p, t.d, t.d_cost, t.d_edges, t.d_firstv);
  hipLaunchKernelGGL(tab_totals_kernel, dim3((unsigned)std::min<long long>(((long long)n + 255) / 256, 1024)), dim3(256), 0, 0, n, t.d_edges, t.d_cost, totals.p);
  t.n = (size_t)n;
  {
    unsigned long long h_tot[3] = {0, 0, 0};
    HIP_TRY(hipMemcpy(h_tot, totals.p, 24, hipMemcpyDeviceToHost));
    t.total_edges = h_tot[0];
    t.total_cost = h_tot[1];
    t.n_with_cost = (size_t)h_tot[2];
  }
  setup_trace("table: expand + totals");
  // dequeue orders: stable descending radix sorts of (key, chunk id)
  {
    const unsigned long long heavy = 2ull * (t.total_cost / (unsigned long long)n) + 1ull;
    unsigned long long *const cost_p = t.d_cost;
    DevBuf<unsigned long long> key0, key1, keyo;
    DevBuf<int> iota;
    HIP_TRY(key0.alloc((size_t)n));
    HIP_TRY(key1.alloc((size_t)n));
    HIP_TRY(keyo.alloc((size_t)n));
    HIP_TRY(iota.alloc((size_t)n));
    hipLaunchKernelGGL(tab_orderkeys_kernel, blocks(n), dim3(256), 0, 0, n, cost_p, heavy, sym_table ? 1 : 0, (!sym_table && g->topo_state == 1) ? 1 : 0,
                       key0.p, key1.p, iota.p);
    for (int m = 0; m < 2; ++m) {
      HIP_TRY(dev_malloc(&t.d_order[m], sizeof(int) * (size_t)n));
      size_t bytes = 0;
      const unsigned long long *keys = m == 0 ? key0.p : key1.p;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, keys, keyo.p, iota.p, t.d_order[m], n));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp.buf.p, bytes, keys, keyo.p, iota.p, t.d_order[m], n));
    }
  }
  setup_trace("table: dequeue orders");
  if (t.allow_split && t.bitmap_min_deg != 0x7fffffff) {  // (0x7fffffff: a table whose kernels never probe a bitmap -- no select pass for nothing)
    SetupTimer bm_timer;
    // Hub rows (longer than the LDS stage, cut into SPLIT chunks) get a dense bitmap over the vertex ids, the longest rows
    // first, within a memory budget (and a quarter of the free device memory): one probe then replaces a ~17-step bisection
    // in HBM. The set is cached on the graph and shared by its tables.
    BitmapSet *bs = nullptr;
    for (auto &b : g->bitmap_sets)
      if (b.min_deg == t.bitmap_min_deg && b.rf == t.rf) bs = &b;
    if (!bs) {
      BitmapSet nb_set;
      nb_set.min_deg = t.bitmap_min_deg;
      nb_set.rf = t.rf;
      const unsigned long long words = ((unsigned long long)nv + 31ull) / 32ull;
      DevBuf<int> flag, iota, sel, nsel;
      HIP_TRY(flag.alloc((size_t)nv));
      HIP_TRY(iota.alloc((size_t)nv));
      HIP_TRY(sel.alloc((size_t)nv));
      HIP_TRY(nsel.alloc(1));
      hipLaunchKernelGGL(tab_hubflag_kernel, blocks(nv), dim3(256), 0, 0, q, t.bitmap_min_deg, g->d_rp, flag.p, iota.p);
      size_t bytes = 0;
      HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceSelect::Flagged(tmp.buf.p, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
      int m = 0;
      HIP_TRY(hipMemcpy(&m, nsel.p, sizeof(int), hipMemcpyDeviceToHost));
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const unsigned long long budget = std::min<unsigned long long>(kBitmapBudget, free_b / 4);
      const size_t nb_max = words ? (size_t)(budget / (words * 4ull)) : 0;
      if (m > 0 && nb_max > 0) {
        DevBuf<int> degs;
        HIP_TRY(degs.alloc((size_t)m));
        hipLaunchKernelGGL(gather_deg_kernel, blocks(m), dim3(256), 0, 0, m, sel.p, g->d_rp, degs.p);
        std::vector<int> hv((size_t)m), hd((size_t)m);
        HIP_TRY(copy_to_host(hv.data(), sel.p, sizeof(int) * (size_t)m));
        HIP_TRY(copy_to_host(hd.data(), degs.p, sizeof(int) * (size_t)m));
        std::vector<int> idx((size_t)m);  // (hub rows only: thousands at most)
        for (int i = 0; i < m; ++i) idx[(size_t)i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a_, int b_) { return hd[(size_t)a_] > hd[(size_t)b_]; });
        const size_t nb = std::min<size_t>((size_t)m, nb_max);
        std::vector<int> rows(nb);
        for (size_t i = 0; i < nb; ++i) rows[i] = hv[(size_t)idx[i]];
        DevBuf<int> d_rows;
        DevBuf<int> row_slot;
        DevBuf<unsigned> bitmaps;
        HIP_TRY(d_rows.alloc(nb));
        HIP_TRY(copy_to_device(d_rows.p, rows.data(), sizeof(int) * nb));
        HIP_TRY(row_slot.alloc((size_t)nv, true));  // (published to the bitmap set below)
        HIP_TRY(hipMemsetAsync(row_slot.p, 0xff, sizeof(int) * (size_t)nv, 0));  // -1
        hipLaunchKernelGGL(tab_rowslot_kernel, blocks((long long)nb), dim3(256), 0, 0, (int)nb, d_rows.p, row_slot.p);
        HIP_TRY(bitmaps.alloc((size_t)nb * (size_t)words, true));
        HIP_TRY(hipMemsetAsync(bitmaps.p, 0, (size_t)nb * (size_t)words * 4, 0));
        hipLaunchKernelGGL(bitmap_build_kernel, dim3((unsigned)nb), dim3(256), 0, 0, g->d_rp, g->d_col, d_rows.p, bitmaps.p, words);
        HIP_TRY(hipDeviceSynchronize());  // (the set is published only after its kernels have succeeded)
        nb_set.d_bitmaps = bitmaps.release();
        nb_set.d_row_slot = row_slot.release();
        nb_set.n = nb;
        nb_set.words = words;
      }
      g->bitmap_sets.push_back(nb_set);
      bs = &g->bitmap_sets.back();
    }
    if (bs->n > 0) {
      t.own_bitmaps = false;
      t.d_bitmaps = bs->d_bitmaps;
      t.d_row_slot = bs->d_row_slot;
      t.n_bitmaps = bs->n;
      t.bitmap_words = bs->words;
      HIP_TRY(dev_malloc(&t.d_slot, sizeof(int) * (size_t)n));
      hipLaunchKernelGGL(tab_chunkslot_kernel, blocks(n), dim3(256), 0, 0, n, t.d, g->d_rp, t.stage_cap, t.d_row_slot, t.d_slot);
    }
    bitmap_ms = bm_timer.ms();
  }
  setup_trace("table: bitmaps");
  HIP_TRY(hipGetLastError());  // (the setup kernels above are launched unchecked)
  HIP_TRY(hipDeviceSynchronize());
  return GM_OK;
}

// the host views of a device-built table (gm_host.h ChunkTable), fetched when a caller first needs them
int table_host_views(gm_graph *g, ChunkTable *t) {
  std::lock_guard<std::mutex> lk(g->mu);
  if (t->host_ready) return GM_OK;
  HIP_TRY(hipSetDevice(g->device));
  const size_t n = t->n;
  t->edge_prefix.assign(n + 1, 0ull);
  t->first_vertex.resize(n);
  t->cost.resize(n);
  if (n) {
    std::vector<int> h_edges(n);
    HIP_TRY(copy_to_host(h_edges.data(), t->d_edges, sizeof(int) * n));
    HIP_TRY(copy_to_host(t->first_vertex.data(), t->d_firstv, sizeof(int) * n));
    HIP_TRY(copy_to_host(t->cost.data(), t->d_cost, sizeof(unsigned long long) * n));
    for (size_t i = 0; i < n; ++i) t->edge_prefix[i + 1] = t->edge_prefix[i] + (unsigned long long)h_edges[i];
    for (int m = 0; m < 2; ++m)
      if (t->d_order[m]) {
        t->order[m].resize(n);
        HIP_TRY(copy_to_host(t->order[m].data(), t->d_order[m], sizeof(int) * n));
      }
  }
  t->host_ready = true;
  return GM_OK;
}

// ---- rank shares with whole chunks (see ShareOrder, gm_host.h) -------------------------------------------------------------------
// Position pos of the order holds record order[pos]; the parts of a chunk are consecutive records of equal cost, and the stable sorts
// behind the dequeue orders keep them consecutive and in part order.  head[pos] = the record is part 0; the chunk's ordinal in the
// order = (inclusive scan of head)[pos] - 1; round robin: ordinal mod world == rank; range: ordinal in the rank's run of the chunks.
__global__ __launch_bounds__(256) void share_head_kernel(const int n, const ChunkRec *__restrict__ recs, const int *__restrict__ order, int *__restrict__ head) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos < n) head[pos] = recs[order ? order[pos] : pos].part == 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void share_flag_kernel(const int n, const int *__restrict__ head, const int *__restrict__ ord_excl, const int world,
                                                         const int rank, const long long lo, const long long hi, int *__restrict__ flag) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n) return;
  const long long ordinal = (long long)ord_excl[pos] + head[pos] - 1;  // the chunk this record is a part of
  flag[pos] = lo < 0 ? (ordinal % world == rank ? 1 : 0) : (ordinal >= lo && ordinal < hi ? 1 : 0);
}
__global__ __launch_bounds__(256) void share_emit_kernel(const int n, const int *__restrict__ order, const int *__restrict__ flag, const int *__restrict__ off,
                                                         const int *__restrict__ edges, int *__restrict__ out, unsigned long long *__restrict__ edge_sum) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long e = 0;
  if (pos < n && flag[pos]) {
    const int rec = order ? order[pos] : pos;
    out[off[pos]] = rec;
    e = (unsigned long long)edges[rec];
  }
  e = gm::wave_sum_u64(e);
  if ((threadIdx.x & 63) == 0 && e) atomicAdd(edge_sum, e);
}

int get_share_order(gm_graph *g, ChunkTable *t, int world, int rank, int policy, int which, const ShareOrder **out) {
  std::lock_guard<std::mutex> lk(g->mu);
  for (const auto &sh : t->shares)
    if (sh.world == world && sh.rank == rank && sh.policy == policy && sh.which == which) { *out = &sh; return GM_OK; }
  if (!t->d_edges) return GM_ERR_UNSUPPORTED;  // (a host-built table: GM_HOST_TABLES devel builds)
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const int n = (int)t->n;
  ShareOrder sh;
  sh.world = world; sh.rank = rank; sh.policy = policy; sh.which = which;
  if (n > 0) {
    PoolScope pool(g);
    ScanTemp tmp;
    DevBuf<int> head, ord, flag, off;
    DevBuf<unsigned long long> esum;
    HIP_TRY(head.alloc((size_t)n + 1));
    HIP_TRY(ord.alloc((size_t)n + 1));
    HIP_TRY(flag.alloc((size_t)n + 1));
    HIP_TRY(off.alloc((size_t)n + 1));
    HIP_TRY(esum.alloc(1));
    HIP_TRY(hipMemsetAsync(head.p, 0, sizeof(int) * ((size_t)n + 1), 0));
    HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int) * ((size_t)n + 1), 0));
    HIP_TRY(hipMemsetAsync(esum.p, 0, sizeof(unsigned long long), 0));
    const int *order = which >= 0 ? t->d_order[which] : nullptr;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(share_head_kernel, grid, block, 0, 0, n, t->d, order, head.p);
    HIP_TRY(dev_exclusive_sum(tmp, head.p, ord.p, (size_t)n + 1));
    long long lo = -1, hi = -1;
    if (policy != GM_PART_ROUND_ROBIN) {  // a contiguous run of the CHUNKS (not of the records)
      int nchunks = 0;
      HIP_TRY(hipMemcpy(&nchunks, ord.p + n, sizeof(int), hipMemcpyDeviceToHost));
      lo = (long long)nchunks * rank / world;
      hi = (long long)nchunks * (rank + 1) / world;
    }
    hipLaunchKernelGGL(share_flag_kernel, grid, block, 0, 0, n, head.p, ord.p, world, rank, lo, hi, flag.p);
    HIP_TRY(dev_exclusive_sum(tmp, flag.p, off.p, (size_t)n + 1));
    int cnt = 0;
    HIP_TRY(hipMemcpy(&cnt, off.p + n, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(dev_malloc(&sh.d, sizeof(int) * (size_t)std::max(cnt, 1)));
    hipLaunchKernelGGL(share_emit_kernel, grid, block, 0, 0, n, order, flag.p, off.p, t->d_edges, sh.d, esum.p);
    if (hipMemcpy(&sh.edges, esum.p, sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) {
      dev_free(sh.d);
      return hip_fail(hipGetLastError(), "share order", __FILE__, __LINE__);
    }
    sh.n = cnt;
  }
  t->shares.push_back(sh);
  *out = &t->shares.back();
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

int get_table(gm_graph *g, int target, bool allow_split, int bit_words, unsigned long long part_cap, int stage_cap,
              ChunkTable **out, const RowFilter &rf, int bitmap_min_deg) {
  std::lock_guard<std::mutex> lk(g->mu);
  for (auto &t : g->tables)
    if (t.target == target && t.allow_split == allow_split && t.bit_words == bit_words && t.part_cap == part_cap && t.stage_cap == stage_cap &&
        t.rf == rf && t.bitmap_min_deg == bitmap_min_deg) { *out = &t; return GM_OK; }
  SetupTimer timer;
  double bitmap_ms = 0;
  const bool sym_table = stage_cap >= kStageCapWide;  // a table of the symmetric-graph patterns (general, mid or big class)
  std::vector<ChunkRec> recs;
  ChunkTable t;
  t.target = target;
  t.allow_split = allow_split;
  t.bit_words = bit_words;
  t.part_cap = part_cap;
  t.stage_cap = stage_cap;
  t.rf = rf;
  t.bitmap_min_deg = bitmap_min_deg;
  if (!gm_sweep_env("GM_HOST_TABLES")) {  // (GM_HOST_TABLES: the same walk in a host loop over the vertices, kept for A/B)
    HIP_TRY(hipSetDevice(g->device));
    PoolScope pool(g);
    int rc = build_table_device(g, t, sym_table, bitmap_ms);
    if (rc) { free_table(t); return rc; }  // (a partially built table owns device memory: out-of-memory on a large graph must not leak it)
    if (gm_sweep_env("GM_TABLE_INFO"))
      fprintf(stderr, "[table/device] stage_cap %d rows (%d,%d] skip (%d,%d]: %zu chunks, est. keys %.3e, %zu bitmaps, edges %llu, %.2f ms\n",
              stage_cap, rf.only_lo, rf.only_hi, rf.skip_lo, rf.skip_hi, t.n, (double)t.total_cost, t.n_bitmaps, t.total_edges, timer.ms());
    g->setup.bitmap_ms += bitmap_ms;
    g->setup.table_ms += timer.ms() - bitmap_ms;
    g->tables.push_back(std::move(t));
    *out = &g->tables.back();
    return GM_OK;
  }
  {
    int rc = host_rp(g, nullptr);
    if (rc) return rc;
  }
  build_chunks(g->h_rp, g->nv, target, allow_split, bit_words, stage_cap, recs, t.max_bit_words, rf);
  // estimated work per chunk (device), then: cut the heavy ones into parts, and fix the dequeue orders
  std::vector<unsigned long long> cost(recs.size());
  if (!recs.empty()) {
    ChunkRec *d_tmp = nullptr;
    unsigned long long *d_cost = nullptr;
    hipError_t e = dev_malloc(&d_tmp, sizeof(ChunkRec) * recs.size());
    if (e == hipSuccess) e = dev_malloc(&d_cost, sizeof(unsigned long long) * recs.size());
    if (e == hipSuccess) e = hipMemcpy(d_tmp, recs.data(), sizeof(ChunkRec) * recs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cost, 0, sizeof(unsigned long long) * recs.size());
    if (e == hipSuccess) {
      hipLaunchKernelGGL(chunk_cost_kernel, dim3((unsigned)recs.size()), dim3(256), 0, 0, g->d_rp, g->d_col, d_tmp, d_cost,
                         sym_table ? 1 : 0, kStageCapWide);
      e = hipMemcpy(cost.data(), d_cost, sizeof(unsigned long long) * recs.size(), hipMemcpyDeviceToHost);
    }
    if (d_tmp) dev_free(d_tmp);
    if (d_cost) dev_free(d_cost);
    if (e != hipSuccess) return hip_fail(e, "chunk_cost_kernel", __FILE__, __LINE__);
  }
  if ((allow_split || sym_table) && part_cap != 0) {  // (clique chunks are never cut: their second phase needs the whole bit-matrix; part_cap 0: never)
    const unsigned long long cap = std::max<unsigned long long>(part_cap, 1);
    std::vector<ChunkRec> cut;
    std::vector<unsigned long long> cut_cost;
    cut.reserve(recs.size());
    cut_cost.reserve(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
      const int bsz = batch_edges(recs[i], g->h_rp, stage_cap, cost[i]);
      const int batches = (recs[i].e_end - recs[i].e_begin + bsz - 1) / bsz;
      // a part must keep every wave of its workgroup busy for several batches: the 16-wave class takes batches of 4 edges
      // (its rows' partners are thousands of keys long) and parts of >= 64 batches, the 4-wave classes parts of >= 16
      const int min_batches = stage_cap == kStageCapBig ? 64 : (stage_cap == kStageCapMid ? 32 : 1);
      const int np = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)std::max(batches / min_batches, 1), cost[i] / cap + (cost[i] % cap != 0)));
      for (int q = 0; q < np; ++q) {
        ChunkRec r = recs[i];
        r.part = q;
        r.nparts = np;
        r.batch = bsz;
        cut.push_back(r);
        cut_cost.push_back(cost[i] / (unsigned long long)np);
      }
    }
    recs.swap(cut);
    cost.swap(cut_cost);
  }
  t.cost = cost;
  t.n = recs.size();
  t.first_vertex.resize(t.n);
  for (size_t i = 0; i < t.n; ++i) t.first_vertex[i] = recs[i].u_begin;
  t.edge_prefix.resize(t.n + 1);
  t.edge_prefix[0] = 0;
  for (size_t i = 0; i < t.n; ++i) {  // task edges of a part = the entries of its batches
    const int nel = recs[i].e_end - recs[i].e_begin, np = recs[i].nparts;
    unsigned long long mine = 0;
    const int bsz = recs[i].batch;
    for (int b = recs[i].part; b * bsz < nel; b += np) mine += (unsigned long long)std::min(bsz, nel - b * bsz);
    t.edge_prefix[i + 1] = t.edge_prefix[i] + mine;
  }
  t.total_edges = t.edge_prefix.back();
  t.total_cost = 0;
  for (auto c : t.cost) t.total_cost += c;
  t.host_ready = true;
  HIP_TRY(dev_malloc(&t.d, sizeof(ChunkRec) * std::max<size_t>(t.n, 1)));
  if (t.n) HIP_TRY(hipMemcpy(t.d, recs.data(), sizeof(ChunkRec) * t.n, hipMemcpyHostToDevice));
  if (t.n) {
    // Longest-processing-time-first dequeue order: the dynamic queue then ends on light chunks, so the tail of a launch
    // (which does not shrink with the number of ranks) stays short; rank r of n owns every n-th entry of this order.
    unsigned long long total_cost = 0;
    for (size_t i = 0; i < t.n; ++i) total_cost += cost[i];
    const unsigned long long heavy = 2ull * (total_cost / t.n) + 1ull;
    for (int m = 0; m < 2; ++m) {
      std::vector<int> &o = t.order[m];
      o.resize(t.n);
      for (size_t i = 0; i < t.n; ++i) o[i] = (int)i;
      // order 0 of the symmetric-graph tables keeps chunk-id order INSIDE the heavy class as well: the heavy chunks are the
      // SPLIT chunks of the hub rows, and consecutive chunks of one row probe the same bitmap -- run together they keep it
      // in L2 (R-MAT-22 diamond: by cost 47.8 ms, by id 39.5 ms)
      const bool classes_only = (m == 0) && sym_table;
      std::stable_sort(o.begin(), o.end(), [&](int a, int b) {
        unsigned long long ca = cost[(size_t)a], cb = cost[(size_t)b];
        if (m == 0) { ca = ca >= heavy ? (classes_only ? 1ull : ca) : 0ull; cb = cb >= heavy ? (classes_only ? 1ull : cb) : 0ull; }
        return ca > cb;
      });
      HIP_TRY(dev_malloc(&t.d_order[m], sizeof(int) * t.n));
      HIP_TRY(hipMemcpy(t.d_order[m], o.data(), sizeof(int) * t.n, hipMemcpyHostToDevice));
    }
  }
  if (allow_split) {
    SetupTimer bm_timer;
    // Hub rows (longer than the LDS stage, cut into SPLIT chunks) get a dense bitmap over the vertex ids, the
    // longest rows first, within a memory budget: one probe then replaces a ~17-step bisection in HBM.
    const unsigned long long words = ((unsigned long long)g->nv + 31ull) / 32ull;
    const unsigned long long budget_bytes = kBitmapBudget;
    std::vector<std::pair<int, int>> big;  // (degree, vertex)
    for (int v = 0; v < g->nv; ++v) {
      const int d = g->h_rp[v + 1] - g->h_rp[v];
      if (d > bitmap_min_deg && !rf.skips(d)) big.push_back({d, v});
    }
    std::sort(big.begin(), big.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first > y.first; });
    const size_t nb = words ? std::min<size_t>(big.size(), (size_t)(budget_bytes / (words * 4ull))) : 0;
    if (nb > 0) {
      std::vector<int> slot_of_row;  // sparse map through a sorted vector
      std::vector<std::pair<int, int>> row_slot(nb);
      std::vector<int> rows(nb);
      for (size_t i = 0; i < nb; ++i) { row_slot[i] = {big[i].second, (int)i}; rows[i] = big[i].second; }
      std::sort(row_slot.begin(), row_slot.end());
      std::vector<int> slots(t.n, -1);
      for (size_t c = 0; c < t.n; ++c) {
        const ChunkRec &r = recs[c];
        if (r.u_end == r.u_begin + 1 && (g->h_rp[r.u_begin + 1] - g->h_rp[r.u_begin]) > stage_cap) {
          auto it = std::lower_bound(row_slot.begin(), row_slot.end(), std::make_pair(r.u_begin, -1));
          if (it != row_slot.end() && it->first == r.u_begin) slots[c] = it->second;
        }
      }
      HIP_TRY(dev_malloc(&t.d_slot, sizeof(int) * t.n));
      HIP_TRY(hipMemcpy(t.d_slot, slots.data(), sizeof(int) * t.n, hipMemcpyHostToDevice));
      {
        std::vector<int> by_vertex((size_t)g->nv, -1);
        for (size_t i = 0; i < nb; ++i) by_vertex[(size_t)rows[i]] = (int)i;
        HIP_TRY(dev_malloc(&t.d_row_slot, sizeof(int) * (size_t)g->nv));
        HIP_TRY(hipMemcpy(t.d_row_slot, by_vertex.data(), sizeof(int) * (size_t)g->nv, hipMemcpyHostToDevice));
      }
      HIP_TRY(dev_malloc(&t.d_bitmaps, (size_t)nb * (size_t)words * 4));
      HIP_TRY(hipMemset(t.d_bitmaps, 0, (size_t)nb * (size_t)words * 4));
      int *d_rows = nullptr;
      HIP_TRY(dev_malloc(&d_rows, sizeof(int) * nb));
      HIP_TRY(hipMemcpy(d_rows, rows.data(), sizeof(int) * nb, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(bitmap_build_kernel, dim3((unsigned)nb), dim3(256), 0, 0, g->d_rp, g->d_col, d_rows, t.d_bitmaps, words);
      hipError_t e = hipDeviceSynchronize();
      dev_free(d_rows);
      if (e != hipSuccess) return hip_fail(e, "bitmap_build_kernel", __FILE__, __LINE__);
      t.n_bitmaps = nb;
      t.bitmap_words = words;
    }
    bitmap_ms = bm_timer.ms();
  }
  HIP_TRY(hipDeviceSynchronize());
  if (gm_sweep_env("GM_TABLE_INFO")) {  // diagnostics
    unsigned long long tc = 0, mx = 0;
    for (auto c : t.cost) { tc += c; mx = std::max(mx, c); }
    fprintf(stderr, "[table] stage_cap %d rows (%d,%d] skip (%d,%d]: %zu chunks, est. keys %.3e (max chunk %.3e), %zu bitmaps, edges %llu\n", stage_cap,
            rf.only_lo, rf.only_hi, rf.skip_lo, rf.skip_hi, t.n, (double)tc, (double)mx, t.n_bitmaps, t.edge_prefix.empty() ? 0ull : t.edge_prefix.back());
  }
  g->setup.bitmap_ms += bitmap_ms;
  g->setup.table_ms += timer.ms() - bitmap_ms;
  g->tables.push_back(std::move(t));
  *out = &g->tables.back();
  return GM_OK;
}

// Edge descriptors (GraphView::edesc): one gather pass over the CSR, once per graph -- the device-side counterpart of
// Graph::init_edgelist (src/common/graph.cc:297-326), which builds the reference's COO task list serially on the host.
__global__ __launch_bounds__(256) void edesc_kernel(long long ne, const int *__restrict__ rp, const int *__restrict__ col, int2 *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) {
    const int v = col[e];
    const int r = rp[v];
    out[e] = make_int2(r, rp[v + 1] - r);
  }
}

// ---- task lists of gm_tch.hip / gm_tch.hip: every edge u -> v of the DAG is a task of ONE of its endpoints (ensure_tasklists, below) ------
// ---- task-major copies of the short lists (gm_host.h: d_colk / d_tdesck) -- for the handles that cannot have the key stream -----------
#ifndef GM_TC_INLINE_MAX_DEFAULT
#define GM_TC_INLINE_MAX_DEFAULT 32
#endif
// len[t] = keys of task t that go into the copies (0: none)
__global__ __launch_bounds__(256) void inl_len_kernel(long long nt, const int2 *__restrict__ tdesc, int lmax, unsigned long long *__restrict__ len) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t <= nt; t += stride) {
    const int n = t < nt ? tdesc[t].y : 0;
    len[t] = (n > 0 && n <= lmax) ? (unsigned long long)n : 0ull;
  }
}
// eight lanes per task: its keys move from their row of col[] to their place in task order, the descriptor follows
__global__ __launch_bounds__(256) void inl_copy_kernel(long long nt, const int2 *__restrict__ tdesc, const unsigned long long *__restrict__ off, int lmax,
                                                        long long ne, int *__restrict__ colk, int2 *__restrict__ tdesck) {
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 3;
  const int sub = threadIdx.x & 7;
  for (long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; t < nt; t += stride) {
    const int2 d = tdesc[t];
    if (d.y > 0 && d.y <= lmax) {
      const long long o = ne + (long long)off[t];
      for (int i = sub; i < d.y; i += 8) colk[o + i] = colk[d.x + i];  // (the first ne entries of colk are col itself)
      if (sub == 0) tdesck[t] = make_int2((int)o, d.y);
    } else if (sub == 0) {
      tdesck[t] = d;
    }
  }
}
// after d_tdesc: failures here leave the handle without copies (the kernels then stream the rows themselves)
static void build_inline_copies(gm_graph *g, ScanTemp &tmp) {
  int lmax = GM_TC_INLINE_MAX_DEFAULT;
  if (const char *e = gm_sweep_env("GM_TC_INLINE_MAX")) lmax = atoi(e);  // (sweeps; 0: no copies)
  if (lmax <= 0 || g->ne <= 0 || g->d_tdesc == nullptr) return;
  const long long nt = g->ne;
  unsigned long long *off = nullptr;  // (lengths in, offsets out: the scan runs in place)
  int *colk = nullptr;
  int2 *tdk = nullptr;
  auto fail = [&]() {
    (void)hipGetLastError();
    for (void *p : {(void *)off, (void *)colk, (void *)tdk})
      if (p) dev_free(p);
  };
  if (dev_malloc(&off, 8 * (size_t)(nt + 1)) != hipSuccess) return fail();
  const long long blocks = std::min<long long>((nt + 256) / 256, (long long)g->cu_count * 32);
  unsigned long long total = 0;
  for (;;) {  // the copies share the 32-bit index space of col[]: halve the limit until they fit
    hipLaunchKernelGGL(inl_len_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, nt, g->d_tdesc, lmax, off);
    if (dev_exclusive_sum(tmp, off, off, (size_t)nt + 1) != hipSuccess) return fail();
    if (hipMemcpy(&total, off + nt, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail();
    if ((unsigned long long)nt + total < 0x7fffff00ull) break;
    lmax >>= 1;
    if (lmax < 4) return fail();
  }
  if (total == 0) return fail();
  if (dev_malloc(&tdk, sizeof(int2) * (size_t)nt) != hipSuccess) return fail();
  if (dev_malloc(&colk, 4 * ((size_t)nt + (size_t)total)) != hipSuccess) return fail();
  if (hipMemcpyAsync(colk, g->d_col, 4 * (size_t)nt, hipMemcpyDeviceToDevice, 0) != hipSuccess) return fail();
  const long long cblocks = std::min<long long>((nt * 8 + 255) / 256, (long long)g->cu_count * 64);
  hipLaunchKernelGGL(inl_copy_kernel, dim3((unsigned)cblocks), dim3(256), 0, 0, nt, g->d_tdesc, off, lmax, nt, colk, tdk);
  if (hipDeviceSynchronize() != hipSuccess) return fail();
  dev_free(off);
  g->d_colk = colk;
  g->d_tdesck = tdk;
  g->n_inline_keys = total;
}

// The task lists by COUNTING PLACEMENT (round 4; until then: one (host, entry) key per edge found by a bisection of the offsets, a radix
// sort of 40 - 260 M key / value pairs, a descriptor pass): eight lanes walk a row; pass 1 counts the tasks of every host, a scan gives
// the offsets, pass 2 repeats the walk and drops every task at its host's cursor -- descriptor, own entry and the low 8 bits of the host
// (the tag of the key stream) written where they stay.  The partner's {start, length} comes from the edge descriptors (coalesced) instead
// of two random reads of the offsets.  The order of a host's tasks is the order of arrival: nothing depends on it.
// Same-address atomics are what such a placement costs on a skewed graph (a hub of R-MAT-22 hosts 10^5 in-edges: 4.3 + 4.6 ms for the two
// passes, no better than the sort): (i) the tasks a row hosts itself are counted per 8-lane group, one atomic per group and step;
// (ii) on a topologically numbered DAG the hubs are the LAST ids: in-edge tasks of the last kHubWin hosts are counted in an LDS
// histogram per workgroup -- pass 2 reserves a range per (workgroup, hub) with one atomic and ranks its tasks inside it in LDS.
constexpr int kHubWin = 4096;
struct TaskWalk {
  int nv, stage_max, topo, hub0;  // hosts >= hub0 are aggregated in LDS (nv: none)
  const int *rp, *col;
  const int2 *edesc;
  int skip_from = 0x7fffffff;     // the out-edges of the rows >= skip_from are no tasks (the triangle count takes them from the core bitmap, gm_ctc.hip)
};
// calls f(u, i, e, ru, du, dv, tail, u_hosts) for every task edge of the rows this workgroup walks (all 64 lanes of a wave stay together:
// f may use wave ballots; `act` = the lane holds a task)
template <class F>
__device__ __forceinline__ void task_walk(const TaskWalk &w, F f) {
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 3;
  const int sub = threadIdx.x & 7;
  const long long u0 = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) >> 3;  // first row of this wave (8 rows per wave and trip)
  for (long long ub = u0; ub < w.nv; ub += stride) {
    const long long u = ub + ((threadIdx.x & 63) >> 3);
    int ru = 0, du = 0;
    if (u < w.nv) {
      ru = w.rp[u];
      du = w.rp[u + 1] - ru;
      if (du > w.stage_max || u >= w.skip_from) du = 0;  // a row the stage cannot take hosts nothing, and its out-edges stay with the chunked kernel (run_pattern)
    }
    const int dmax = wave_max_nonneg(du);
    for (int i = sub; i - sub < dmax; i += 8) {  // wave-uniform trip count
      const bool act = i < du;
      const int e = ru + (act ? i : 0);
      const int2 dv = act ? w.edesc[e] : make_int2(0, 0);  // {rp[v], d+(v)} of the entry's target v
      // the host = the endpoint whose list is NOT streamed: N+(v) whole, or N+(u) -- under a topological numbering only its part beyond
      // v -- whichever is shorter (ties: the source hosts); a list that does not fit the stage never hosts
      const int tail = w.topo ? du - i - 1 : du;
      const bool u_hosts = dv.y > w.stage_max || tail >= dv.y;
      f(act, (int)u, i, e, ru, du, dv, tail, u_hosts);
    }
  }
}

template <bool PLACE>
__global__ __launch_bounds__(256) void task_rows_kernel(const TaskWalk w, int *__restrict__ cnt /* PLACE: the cursors */, const int *__restrict__ trp,
                                                         int2 *__restrict__ tdesc, int *__restrict__ tedge) {
  __shared__ int hist[kHubWin];
  for (int h = threadIdx.x; h < kHubWin; h += 256) hist[h] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane & ~7;
  auto put = [&](const int slot, const int host, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts) {
    tdesc[slot] = u_hosts ? dv : (w.topo ? make_int2(e + 1, tail) : make_int2(ru, du));
    tedge[slot] = e;
  };
  task_walk(w, [&](const bool act, const int u, const int i, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts) {
    // (i) the tasks the row hosts itself: one atomic per 8-lane group
    const unsigned long long mu = __ballot(act && u_hosts);
    const unsigned gm = (unsigned)(mu >> g8) & 255u;
    int ubase = 0;
    if (gm != 0u && sub == (int)__builtin_ctz(gm)) ubase = atomicAdd(&cnt[u], (int)__builtin_popcount(gm));
    if (PLACE && gm != 0u) {
      ubase = __shfl(ubase, g8 + (int)__builtin_ctz(gm));
      if (act && u_hosts) put(trp[u] + ubase + (int)__builtin_popcount(gm & ((1u << sub) - 1u)), u, e, ru, du, dv, tail, true);
    }
    // (ii) in-edge tasks: the target hosts -- a hub through the LDS histogram, anybody else through its global cursor
    if (act && !u_hosts) {
      const int host = w.col[e];
      if (host >= w.hub0) {
        atomicAdd(&hist[host - w.hub0], 1);
      } else if (!PLACE) {
        atomicAdd(&cnt[host], 1);
      } else {
        put(trp[host] + atomicAdd(&cnt[host], 1), host, e, ru, du, dv, tail, false);
      }
    }
  });
  __syncthreads();
  for (int h = threadIdx.x; h < kHubWin; h += 256) {
    const int c = hist[h];
    if (c) {
      const int b = atomicAdd(&cnt[w.hub0 + h], c);  // pass 1: the count; pass 2: this workgroup's range among the hub's tasks
      if (PLACE) hist[h] = b;  // (the ranks inside the range are taken from the same word)
    }
  }
  if (!PLACE) return;
  __syncthreads();
  task_walk(w, [&](const bool act, const int u, const int i, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts) {
    if (act && !u_hosts) {
      const int host = w.col[e];
      if (host >= w.hub0) {
        const int h = host - w.hub0;
        put(trp[host] + atomicAdd(&hist[h], 1), host, e, ru, du, dv, tail, false);
      }
    }
  });
}

// ---- the key stream of the triangle count, built DIRECTLY (round 4: until then from the task lists above -- every task placed as a 13-byte
// record by scattered stores, three scans over the edges, the records read back to copy their keys: 7.9 of the 18 ms a first triangle count
// of R-MAT-22 took, 43 of 119 on R-MAT-24).  The same two walks with one packed counter per host -- keys of the short lists in the low
// 32 bits, tasks with longer lists in the high 32: pass 1 counts, two scans over the VERTICES give the hosts' offsets, pass 2 reserves a
// task's place with the same atomic and writes the keys (tagged with the host's low byte) or the {start, length} of a longer list where
// they stay.  A group of eight lanes copies the short lists of its eight tasks together, eight keys a step.  The order of a host's keys
// is the order of arrival: the kernel looks every key up on its own.
static bool keystream_possible(const gm_graph *g) {  // ids must leave bits 24..31 to the host tag
  return g->nv <= (1 << 24) && !gm_sweep_env("GM_TC_NO_KEY_STREAM");
}
struct KeyCopy { int src, dst, len, own; unsigned tag; };
// EDGES (the edge supports, gm_sup.hip): beside every key the DAG entry it was copied from and the entry of its task's own edge, beside
// every longer list the entry of its task's own edge -- read only where a key is found
struct KstEdges { int2 *kst_et; int *tedgel; };  // kst_et[k] = {entry of key k, entry of its task's own edge}
template <bool PLACE, int WIN, bool EDGES>
__global__ __launch_bounds__(256) void kst_rows_kernel(const TaskWalk w, const int lmax, unsigned long long *__restrict__ cnt /* PLACE: the cursors */,
                                                        const int *__restrict__ kst_rp, const int *__restrict__ trpl, unsigned *__restrict__ kst,
                                                        int2 *__restrict__ tdescl, const KstEdges ed) {
  __shared__ unsigned long long hist[WIN];
  const int hubwin = w.nv - w.hub0;  // (<= WIN: ensure_keystream)
  for (int h = threadIdx.x; h < hubwin; h += 256) hist[h] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane & ~7;
  // the list a task streams, and its weight in the packed counters
  auto task_list = [&](const bool act, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts, int &start, int &len) {
    start = u_hosts ? dv.x : (w.topo ? e + 1 : ru);
    len = !act ? 0 : (u_hosts ? dv.y : (w.topo ? tail : du));
    return len == 0 ? 0ull : (len <= lmax ? (unsigned long long)len : (1ull << 32));
  };
  auto copy_group = [&](const KeyCopy c) {  // c.len = 0: nothing of this lane's
    for (int j = 0; j < 8; ++j) {
      const int lj = __shfl(c.len, g8 + j), sj = __shfl(c.src, g8 + j), dj = __shfl(c.dst, g8 + j);
      const unsigned tj = (unsigned)__shfl((int)c.tag, g8 + j);
      for (int i = sub; i < lj; i += 8) kst[dj + i] = (unsigned)w.col[sj + i] | tj;
      if (EDGES) {
        const int oj = __shfl(c.own, g8 + j);
        for (int i = sub; i < lj; i += 8) ed.kst_et[dj + i] = make_int2(sj + i, oj);
      }
    }
  };
  auto put = [&](const bool mine, const int host, const unsigned long long pos, const int start, const int len, const int e) {
    KeyCopy c{start, 0, 0, e, (unsigned)(host & 255) << 24};
    if (mine) {
      if (len > lmax) {
        const int slot = trpl[host] + (int)(pos >> 32);
        tdescl[slot] = make_int2(start, len);
        if (EDGES) ed.tedgel[slot] = e;
      } else {
        c.dst = kst_rp[host] + (int)(unsigned)pos;
        c.len = len;
      }
    }
    copy_group(c);
  };
  task_walk(w, [&](const bool act, const int u, const int i, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts) {
    int start, len;
    const unsigned long long wgt = task_list(act, e, ru, du, dv, tail, u_hosts, start, len);
    // (i) the tasks the row hosts itself: one atomic per 8-lane group (an inclusive scan of the packed weights over the group)
    const unsigned long long mine = u_hosts ? wgt : 0ull;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const unsigned lo = (unsigned)__shfl((int)(unsigned)incl, lane - d), hi = (unsigned)__shfl((int)(unsigned)(incl >> 32), lane - d);
      if (sub >= d) incl += ((unsigned long long)hi << 32) | lo;
    }
    const unsigned long long total = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(incl >> 32), g8 + 7) << 32) | (unsigned)__shfl((int)(unsigned)incl, g8 + 7);
    unsigned long long ubase = 0ull;
    if (total != 0ull && sub == 0) ubase = atomicAdd(&cnt[u], total);
    // (ii) in-edge tasks: the target hosts -- a hub through the LDS histogram, anybody else through its global cursor
    int host = u;
    unsigned long long pos = 0ull;
    bool now = mine != 0ull;
    if (!u_hosts && wgt != 0ull) {
      host = w.col[e];
      if (host >= w.hub0) atomicAdd(&hist[host - w.hub0], wgt);
      else { pos = atomicAdd(&cnt[host], wgt); now = true; }
    }
    if (PLACE) {
      if (total != 0ull) {
        ubase = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(ubase >> 32), g8) << 32) | (unsigned)__shfl((int)(unsigned)ubase, g8);
        if (mine != 0ull) pos = ubase + incl - mine;
      }
      put(now, host, pos, start, len, e);
    }
  });
  __syncthreads();
  for (int h = threadIdx.x; h < hubwin; h += 256) {
    const unsigned long long c = hist[h];
    if (c) {
      const unsigned long long b = atomicAdd(&cnt[w.hub0 + h], c);  // pass 1: the count; pass 2: this workgroup's range among the hub's
      if (PLACE) hist[h] = b;
    }
  }
  if (!PLACE || w.hub0 >= w.nv) return;
  __syncthreads();
  task_walk(w, [&](const bool act, const int u, const int i, const int e, const int ru, const int du, const int2 dv, const int tail, const bool u_hosts) {
    int start, len;
    const unsigned long long wgt = task_list(act, e, ru, du, dv, tail, u_hosts, start, len);
    int host = 0;
    unsigned long long pos = 0ull;
    bool now = false;
    if (!u_hosts && wgt != 0ull) {
      host = w.col[e];
      if (host >= w.hub0) { pos = atomicAdd(&hist[host - w.hub0], wgt); now = true; }
    }
    put(now, host, pos, start, len, e);
  });
}
__global__ __launch_bounds__(256) void kst_unpack_kernel(int nv, const unsigned long long *__restrict__ cnt, unsigned long long *__restrict__ keys, int *__restrict__ longs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nv) return;
  const unsigned long long c = v < nv ? cnt[v] : 0ull;
  keys[v] = c & 0xffffffffull;
  longs[v] = (int)(c >> 32);
}
__global__ __launch_bounds__(256) void kst_narrow_kernel(int nv, const unsigned long long *__restrict__ in, int *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v <= nv) out[v] = (int)in[v];
}

// d_kst / d_kst_rp / d_trpl / d_tdescl of a handle (gm_host.h): GM_OK also when the handle cannot have them (ids beyond 24 bits, keys
// beyond the 32-bit index space at every list limit) -- *built says which
// edges: with the entries the edge supports need (gm_host.h KeyStream): a handle whose stream was built without them gets a SECOND set of
// keys / longer lists with them -- same offsets (the counts per host do not depend on the order of arrival), its own order.
// the hub corner the triangle count takes on the matrix cores (gm_ctc.hip): its size on this handle -- the last h vertices, word-aligned in the
// core bitmap -- or 0.  The masked product costs h^3 / 6 bit-products whatever the corner holds, the streamed kernel ~rho^2 h^3 / 8 keys for
// a corner of density rho (|E| = rho h^2 / 2 edges, lists of rho h / 2 keys, half of a list streamed), so the corner pays from a density
// on: measured (profiles/r05/ab_tc_core.txt) R-MAT-22 at h = 8 K / 16 K / 32 K (11 / 4.5 / 2.0 %): 1.96 / 1.89 / 3.09 ms against 2.47
// without, R-MAT-24 at 16 K / 32 K (12.7 / 5.4 %): 26.1 / 24.5 against 32.2 -- the largest power-of-two fraction of the core bitmap whose
// density is >= kTcCoreMinDensity.  GM_TC_CORE_H = 0 switches the corner off, any other value forces that size (tests: small graphs).
constexpr double kTcCoreMinDensity = 0.03;
static int tc_core_size(gm_graph *g, const char *env = "GM_TC_CORE_H", bool blocks_only = false) {
  long long want = kTcCoreHDefault;
  bool forced = false;
  if (const char *e = gm_opt(env)) { want = atoll(e); forced = true; }
  if (want <= 0 || g->nv < 64) return 0;
  if (ensure_core_bitmap(g) != GM_OK || g->core_state != 1) return 0;  // (not topologically numbered, no room: everything through the stream)
  const long long cap = std::min<long long>((long long)g->core_h, (long long)kCtcMaxH);
  auto aligned = [&](long long h) {  // the corner starts at a word of the bitmap's rows
    const long long off = ((long long)g->core_h - h + 31) / 32 * 32;
    return off >= g->core_h ? 0ll : (long long)g->core_h - off;
  };
  if (forced) {
    const long long h = aligned(std::min(want, cap));
    return (int)((blocks_only && h % 512 != 0) ? 0 : h);  // (the supports' corner runs on the block kernel only)
  }
  // a corner is worth its MFMA pass where the hubs are a small part of the graph: at most a quarter of the vertices, at least 1024 of them
  for (long long h = cap; h >= 1024; h >>= 1) {
    if ((long long)g->nv < 4 * h) continue;
    int e0 = 0;
    if (hipMemcpy(&e0, g->d_rp + (g->nv - h), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const double rho = (double)(g->ne - (long long)e0) / (0.5 * (double)h * (double)h);
    if (rho >= kTcCoreMinDensity && !(blocks_only && aligned(h) % 512 != 0)) return (int)aligned(h);
  }
  return 0;
}
int ensure_keystream(gm_graph *g, bool edges, bool *built, bool allow_core) {
  auto have = [&]() { return g->d_kst_rp != nullptr && (!edges || g->d_kst_et != nullptr); };
  *built = have();
  if (*built || g->kst_state == 2 || g->ne == 0) return GM_OK;
  // (a stream without the hub corner's rows cannot carry the edge supports: they take the task lists)
  if (edges && g->d_kst_rp != nullptr && g->kst_skip_from < g->nv) return GM_OK;
  {
    const int rc = ensure_edesc(g);  // (takes the lock itself)
    if (rc) return rc;
  }
  const int tc_h = (allow_core && !edges && g->d_kst_rp == nullptr) ? tc_core_size(g) : 0;  // (takes the lock itself)
  std::lock_guard<std::mutex> lk(g->mu);
  *built = have();
  if (*built || g->kst_state == 2) return GM_OK;
  if (edges && g->d_kst_rp != nullptr && g->kst_skip_from < g->nv) return GM_OK;
  const bool second = g->d_kst_rp != nullptr;  // the offsets exist: only the place pass, into the second set
  int lmax = second ? g->kst_lmax : GM_TC_INLINE_MAX_DEFAULT;
  if (!second) {
    if (const char *e = gm_sweep_env("GM_TC_INLINE_MAX")) lmax = atoi(e);  // (sweeps)
    if (!keystream_possible(g) || lmax < 4) {
      g->kst_state = 2;
      return GM_OK;
    }
  }
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const size_t nv1 = (size_t)g->nv + 1;
  PoolScope pool(g);
  ScanTemp tmp;
  bool topo = false;
  {
    const int rc = graph_is_topological(g, &topo);
    if (rc) return rc;
    if (gm_sweep_env("GM_TC_NO_TRIM")) topo = false;  // (A/B: whole lists streamed on a topologically numbered DAG too)
  }
  DevBuf<unsigned long long> cnt, keys, keyoff;
  DevBuf<int> longs;
  HIP_TRY(cnt.alloc(nv1));
  HIP_TRY(keys.alloc(nv1));
  HIP_TRY(keyoff.alloc(nv1));
  HIP_TRY(longs.alloc(nv1));
  // (few, fat workgroups where a hub window exists: a workgroup's LDS histogram of the hub hosts pays when it sees many rows)
  // The two passes choose their windows independently (a window only says which hosts are aggregated in LDS): the count pass is all
  // atomics and wants the larger one, the place pass waits for its returning atomics and the lists it copies and wants more workgroups
  // per CU.  Measured, window / workgroups per CU, count + place ms: R-MAT-22 4096 / 8: 1.38 + 2.51, 2048 / 16: 1.43 + 2.16;
  // R-MAT-24 4096 / 8: 8.5 + 14.7, 2048 / 16: 10.1 + 13.6.
  int win_count = kHubWin, win_place = 2048, per_cu_count = topo ? 8 : 64, per_cu_place = topo ? 16 : 64;
  if (const char *e = gm_sweep_env("GM_KST_HUB_WIN")) win_count = win_place = std::max(0, std::min(atoi(e), kHubWin));  // (sweeps)
  if (const char *e = gm_sweep_env("GM_KST_WG_PER_CU")) per_cu_count = per_cu_place = std::max(1, atoi(e));
  const long long blocks_count = std::min<long long>(((long long)g->nv * 8 + 255) / 256, (long long)g->cu_count * per_cu_count);
  const long long blocks_place = std::min<long long>(((long long)g->nv * 8 + 255) / 256, (long long)g->cu_count * per_cu_place);
  TaskWalk tw;
  tw.nv = g->nv; tw.stage_max = kTctStageMax; tw.topo = topo ? 1 : 0;
  tw.rp = g->d_rp; tw.col = g->d_col; tw.edesc = g->d_edesc;
  const int skip_from = (!second && topo && tc_h > 0) ? g->nv - tc_h : 0x7fffffff;
  tw.skip_from = skip_from;
  long long core_edges = 0;
  if (skip_from < g->nv) {
    int e0 = 0;
    HIP_TRY(hipMemcpy(&e0, g->d_rp + skip_from, sizeof(int), hipMemcpyDeviceToHost));
    core_edges = g->ne - (long long)e0;
  }
  TaskWalk tw_count = tw, tw_place = tw;
  tw_count.hub0 = topo ? std::max(0, g->nv - win_count) : g->nv;
  tw_place.hub0 = topo ? std::max(0, g->nv - win_place) : g->nv;
  int *krp = second ? g->d_kst_rp : nullptr, *trpl = second ? g->d_trpl : nullptr;
  unsigned *kst = nullptr;
  int2 *tdl = nullptr;
  KstEdges ed{nullptr, nullptr};
  auto fail = [&](hipError_t e, const char *what) {
    for (void *q : {second ? nullptr : (void *)krp, second ? nullptr : (void *)trpl, (void *)kst, (void *)tdl, (void *)ed.kst_et, (void *)ed.tedgel})
      if (q) dev_free(q);
    return hip_fail(e, what, __FILE__, __LINE__);
  };
  hipError_t e = hipSuccess;
  if (!second && ((e = dev_malloc(&krp, sizeof(int) * nv1)) != hipSuccess || (e = dev_malloc(&trpl, sizeof(int) * nv1)) != hipSuccess)) return fail(e, "hipMalloc(key stream offsets)");
  unsigned long long total = second ? g->n_inline_keys : 0, key_limit = 0x7fffff00ull;
  if (const char *e = gm_opt("GM_KST_MAX_KEYS")) key_limit = std::min<unsigned long long>(key_limit, (unsigned long long)std::max(1ll, atoll(e)));  // (tests)
  int nlong = second ? g->n_long_tasks : 0;
  while (!second) {  // the stream is indexed with 32 bits: halve the limit of a "short" list until it fits
    if ((e = hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long) * nv1, 0)) != hipSuccess) return fail(e, "hipMemsetAsync");
    if (win_count <= 2048) hipLaunchKernelGGL((kst_rows_kernel<false, 2048, false>), dim3((unsigned)blocks_count), dim3(256), 0, 0, tw_count, lmax, cnt.p, nullptr, nullptr, nullptr, nullptr, ed);
    else hipLaunchKernelGGL((kst_rows_kernel<false, kHubWin, false>), dim3((unsigned)blocks_count), dim3(256), 0, 0, tw_count, lmax, cnt.p, nullptr, nullptr, nullptr, nullptr, ed);
    hipLaunchKernelGGL(kst_unpack_kernel, dim3((unsigned)((nv1 + 255) / 256)), dim3(256), 0, 0, g->nv, cnt.p, keys.p, longs.p);
    if ((e = dev_exclusive_sum(tmp, keys.p, keyoff.p, nv1)) != hipSuccess) return fail(e, "ExclusiveSum");
    if ((e = dev_exclusive_sum(tmp, longs.p, trpl, nv1)) != hipSuccess) return fail(e, "ExclusiveSum");
    if ((e = hipMemcpy(&total, keyoff.p + g->nv, 8, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    if ((e = hipMemcpy(&nlong, trpl + g->nv, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    if (total < key_limit) break;
    lmax >>= 1;
    if (lmax < 4) {
      dev_free(krp);
      dev_free(trpl);
      g->kst_state = 2;
      return GM_OK;
    }
  }
  setup_trace("key stream: count pass + scans");
  if (!second) hipLaunchKernelGGL(kst_narrow_kernel, dim3((unsigned)((nv1 + 255) / 256)), dim3(256), 0, 0, g->nv, keyoff.p, krp);
  const size_t nk = (size_t)std::max<unsigned long long>(total, 1), nl = (size_t)std::max(nlong, 1);
  if ((e = dev_malloc(&kst, sizeof(unsigned) * nk)) != hipSuccess) return fail(e, "hipMalloc(key stream)");
  if ((e = dev_malloc(&tdl, sizeof(int2) * nl)) != hipSuccess) return fail(e, "hipMalloc(long lists)");
  if (edges && ((e = dev_malloc(&ed.kst_et, sizeof(int2) * nk)) != hipSuccess || (e = dev_malloc(&ed.tedgel, sizeof(int) * nl)) != hipSuccess))
    return fail(e, "hipMalloc(key stream entries)");
  if ((e = hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long) * nv1, 0)) != hipSuccess) return fail(e, "hipMemsetAsync");
  const dim3 pg((unsigned)blocks_place), pb(256);
  if (edges) {
    if (win_place <= 2048) hipLaunchKernelGGL((kst_rows_kernel<true, 2048, true>), pg, pb, 0, 0, tw_place, lmax, cnt.p, krp, trpl, kst, tdl, ed);
    else hipLaunchKernelGGL((kst_rows_kernel<true, kHubWin, true>), pg, pb, 0, 0, tw_place, lmax, cnt.p, krp, trpl, kst, tdl, ed);
  } else {
    if (win_place <= 2048) hipLaunchKernelGGL((kst_rows_kernel<true, 2048, false>), pg, pb, 0, 0, tw_place, lmax, cnt.p, krp, trpl, kst, tdl, ed);
    else hipLaunchKernelGGL((kst_rows_kernel<true, kHubWin, false>), pg, pb, 0, 0, tw_place, lmax, cnt.p, krp, trpl, kst, tdl, ed);
  }
  if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "kst_rows_kernel");
  setup_trace("key stream: place pass");
  if (second) {  // (the first set stays what the triangle count reads)
    g->d_kst2 = kst;
    g->d_tdescl2 = tdl;
  } else {
    g->d_kst = kst;
    g->d_trpl = trpl;
    g->d_tdescl = tdl;
    g->n_inline_keys = total;
    g->n_long_tasks = nlong;
    g->kst_lmax = lmax;
    if (skip_from < g->nv) {
      g->tc_core_edges = core_edges;
      g->tc_core_h = g->nv - skip_from;
      g->kst_skip_from = skip_from;
    }
  }
  if (edges) {
    g->d_tedgel = ed.tedgel;
    g->d_kst_et = ed.kst_et;
  }
  if (!second) g->d_kst_rp = krp;
  g->setup.table_ms += timer.ms();
  *built = true;
  return GM_OK;
}

int ensure_tasklists(gm_graph *g, bool /*with_edges*/) {
  // The tasks' own entries (tedge, 4 B per edge: what the edge supports need) are always built with the lists.  Round 3 built them on
  // demand by freeing and rebuilding trp / tdesc -- under a launch of another thread that had already copied those pointers.
  if (__atomic_load_n(&g->d_tdesc, __ATOMIC_ACQUIRE) || g->ne == 0) return GM_OK;
  {
    const int rc = ensure_edesc(g);  // (takes the lock itself)
    if (rc) return rc;
  }
  // the hub corner (gm_ctc.hip): its edges' supports / triangles come from the matrix cores -- where every row fits the stage (the rows beyond
  // it are the chunked kernel's, run_pattern) and the corner is a whole number of 512-vertex chunks
  const int tl_h = g->max_deg <= kTctStageMax ? tc_core_size(g, "GM_SUP_CORE_H", true) : 0;  // (takes the lock itself)
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->d_tdesc) return GM_OK;
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const size_t ne = (size_t)g->ne, nv1 = (size_t)g->nv + 1;
  PoolScope pool(g);
  DevBuf<int> cnt;
  ScanTemp tmp;
  bool topo = false;
  {
    const int rc = graph_is_topological(g, &topo);
    if (rc) return rc;
    if (gm_sweep_env("GM_TC_NO_TRIM")) topo = false;  // (A/B: whole lists streamed on a topologically numbered DAG too)
  }
  setup_trace("tasks: sorted / topological check");
  HIP_TRY(cnt.alloc(nv1));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(int) * nv1, 0));
  // (few, fat workgroups: a workgroup's LDS histogram of the hub hosts pays when it sees many rows)
  // (where no hub window exists -- a DAG that is not numbered topologically -- many thin workgroups: power-law LJ-size 30.9 vs 9.8 ms)
  const long long blocks = std::min<long long>(((long long)g->nv * 8 + 255) / 256, (long long)g->cu_count * (topo ? 8 : 64));
  TaskWalk tw;
  tw.nv = g->nv; tw.stage_max = kTctStageMax; tw.topo = topo ? 1 : 0;
  tw.hub0 = topo ? std::max(0, g->nv - kHubWin) : g->nv;
  tw.rp = g->d_rp; tw.col = g->d_col; tw.edesc = g->d_edesc;
  const int tl_skip = (topo && tl_h > 0) ? g->nv - tl_h : 0x7fffffff;
  tw.skip_from = tl_skip;
  hipLaunchKernelGGL((task_rows_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, 0, tw, cnt.p, nullptr, nullptr, nullptr);
  setup_trace("tasks: count pass");
  int *trp = nullptr, *tedge = nullptr;
  int2 *td = nullptr;
  HIP_TRY(dev_malloc(&trp, sizeof(int) * nv1));
  hipError_t e = dev_exclusive_sum(tmp, cnt.p, trp, nv1);
  if (e == hipSuccess) e = dev_malloc(&td, sizeof(int2) * ne);
  if (e == hipSuccess) e = dev_malloc(&tedge, sizeof(int) * ne);
  if (e == hipSuccess) e = hipMemsetAsync(td, 0, sizeof(int2) * ne, 0);  // (the edges of rows beyond the stage are no tasks: empty descriptors at the end)
  if (e == hipSuccess) e = hipMemsetAsync(tedge, 0, sizeof(int) * ne, 0);
  if (e == hipSuccess) e = hipMemsetAsync(cnt.p, 0, sizeof(int) * nv1, 0);
  setup_trace("tasks: scan, allocations, clears");
  if (e == hipSuccess) {
    const long long pblocks = std::min<long long>(((long long)g->nv * 8 + 255) / 256, (long long)g->cu_count * (topo ? 16 : 64));  // (the place pass waits for its returning atomics)
    hipLaunchKernelGGL((task_rows_kernel<true>), dim3((unsigned)pblocks), dim3(256), 0, 0, tw, cnt.p, trp, td, tedge);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {
    dev_free(trp);
    if (td) dev_free(td);
    if (tedge) dev_free(tedge);
    return hip_fail(e, "task lists", __FILE__, __LINE__);
  }
  if (tl_skip < g->nv) {
    g->tl_skip_from = tl_skip;
    g->tl_core_h = g->nv - tl_skip;
  }
  g->d_trp = trp;
  g->d_tedge = tedge;
  // d_tdesc is what the lock-free check at the top reads: published LAST, with release order, so that a first caller on another thread
  // that sees it also sees the corner's skip row and the other arrays
  __atomic_store_n(&g->d_tdesc, td, __ATOMIC_RELEASE);
  setup_trace("tasks: place pass");
  if (!keystream_possible(g)) build_inline_copies(g, tmp);  // (a handle that cannot have the key stream: task-major copies of the short lists)
  setup_trace("tasks: inline copies");
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// ---- MATCH MASKS of the edge supports (gm_sup.hip, round 5).  An IN-EDGE task -- the target v = s_i of u -> v hosts, the tail
// N+(u)[i + 1 ..) is streamed -- finds triangles (u, s_i, s_j) whose "streamed" edge (u, s_j) is an entry of row u BEHIND the task's own
// entry: instead of one memory-side atomic per match the task stores WHICH keys matched -- bit k of its mask = the k-th key of the tail --
// with plain stores, and a second pass sums the masks of a row by column.  87 % of the triangles of an R-MAT graph are found by in-edge
// tasks, two thirds of them by tasks with tails of >= 32 keys.  Here: the size of every entry's mask (64-bit words; 0 = no mask: an
// out-edge task, a tail below `lmin`, a row beyond the stage), their offsets by one scan, and the offsets again in task order.
__global__ __launch_bounds__(256) void sup_mask_size_kernel(const TaskWalk w, const int lmin, unsigned long long *__restrict__ sz) {
  task_walk(w, [&](const bool act, const int, const int, const int e, const int, const int, const int2, const int tail, const bool u_hosts) {
    // (a LONG list -- streamed one task at a time, gm_hset.h -- looks all the tiles of its last group up: kSupTiles - 1 spare words)
    if (act && !u_hosts && tail >= lmin) sz[e] = (unsigned long long)(((tail + 63) >> 6) + (tail >= kLongList ? kSupMaskSpare : 0));
  });
}
__global__ __launch_bounds__(256) void sup_mask_off_kernel(const long long ne, const unsigned long long *__restrict__ off, unsigned *__restrict__ emoff) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (long long)gridDim.x * blockDim.x)
    emoff[e] = off[e + 1] != off[e] ? (unsigned)off[e] : kNoMask;
}
__global__ __launch_bounds__(256) void sup_mask_task_kernel(const long long nt, const int2 *__restrict__ tdesc, const int *__restrict__ tedge,
                                                            const unsigned *__restrict__ emoff, unsigned *__restrict__ tmoff) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += (long long)gridDim.x * blockDim.x) {
    const int2 d = tdesc[t];
    const int e = tedge[t];
    // (an in-edge task streams the entries right behind its own: start == own entry + 1; an out-edge task of the same entry has no mask)
    tmoff[t] = (d.y > 0 && d.x == e + 1) ? emoff[e] : kNoMask;
  }
}

// the rows with tails of more than 64 keys (a second mask word: sup_far_kernel), widest ids first: flag, scan, scatter
__global__ __launch_bounds__(256) void sup_far_flag_kernel(const int nv, const int *__restrict__ rp, const int lmin, const int stage_max, const int skip_from,
                                                           int *__restrict__ flag) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // position k = vertex nv - 1 - k
  if (k >= nv) return;
  const int v = nv - 1 - k, d = rp[v + 1] - rp[v];
  flag[k] = (v < skip_from && d <= stage_max && d - 1 > GM_WAVE && d - 1 >= lmin) ? 1 : 0;
}
__global__ __launch_bounds__(256) void sup_far_list_kernel(const int nv, const int *__restrict__ flag, const int *__restrict__ pos, int *__restrict__ rows) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nv && flag[k]) rows[pos[k]] = nv - 1 - k;
}

// d_csym / d_cfirst: the symmetric bit matrix of the hub corner the task lists leave out, and the positions the supports' epilogue needs
int ensure_sup_corner(gm_graph *g) {
  if (g->tl_skip_from >= g->nv || g->d_csym) return GM_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->d_csym) return GM_OK;
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const int h = g->tl_core_h, words = h / 32;
  const size_t cells = (size_t)h * (size_t)words;
  unsigned *bits = nullptr;
  unsigned short *first_pos = nullptr;
  HIP_TRY(dev_malloc(&bits, cells * sizeof(unsigned)));
  hipError_t e = dev_malloc(&first_pos, cells * sizeof(unsigned short));
  if (e == hipSuccess) e = hipMemsetAsync(bits, 0, cells * sizeof(unsigned), 0);
  if (e == hipSuccess) e = hipMemsetAsync(first_pos, 0, cells * sizeof(unsigned short), 0);
  int e0 = 0;
  if (e == hipSuccess) e = hipMemcpy(&e0, g->d_rp + g->tl_skip_from, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = launch_core_sym_fill(g->nv, g->tl_skip_from, words, (long long)e0, g->ne, g->d_rp, g->d_col, bits, first_pos, g->cu_count, 0);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    dev_free(bits);
    if (first_pos) dev_free(first_pos);
    return hip_fail(e, "symmetric corner", __FILE__, __LINE__);
  }
  g->d_cfirst = first_pos;
  g->d_csym = bits;
  setup_trace("supports: symmetric corner");
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// the shortest tail that gets a mask on this handle: kSupMaskMinTail, or the option GM_SUP_MASK_MIN as it stood when the handle's masks
// were laid out (ensure_sup_masks) -- the kernels must use the value the offsets were built with
int sup_mask_min_tail(const gm_graph *g) { return g->smask_min_tail > 0 ? g->smask_min_tail : kSupMaskMinTail; }

int ensure_sup_masks(gm_graph *g) {
  if (g->smask_state != 0) return GM_OK;
  if (!g->d_tdesc || !g->d_tedge || !g->d_edesc) return GM_ERR_INVALID;  // (after ensure_tasklists)
  bool topo = false;
  {
    const int rc = graph_is_topological(g, &topo);
    if (rc) return rc;
  }
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->smask_state != 0) return GM_OK;
  if (!topo || gm_sweep_env("GM_TC_NO_TRIM") || gm_opt("GM_SUP_NO_MASKS") || g->ne < 1) {  // (tails exist on a topologically numbered DAG only)
    g->smask_state = 2;
    return GM_OK;
  }
  if (const char *e = gm_opt("GM_SUP_MASK_MIN")) g->smask_min_tail = std::max(1, atoi(e));
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const size_t ne = (size_t)g->ne;
  PoolScope pool(g);
  DevBuf<unsigned long long> off;
  ScanTemp tmp;
  HIP_TRY(off.alloc(ne + 1));
  HIP_TRY(hipMemsetAsync(off.p, 0, sizeof(unsigned long long) * (ne + 1), 0));
  TaskWalk tw;
  tw.nv = g->nv; tw.stage_max = kTctStageMax; tw.topo = 1; tw.hub0 = g->nv;
  tw.rp = g->d_rp; tw.col = g->d_col; tw.edesc = g->d_edesc;
  tw.skip_from = g->tl_skip_from;  // (the corner's rows have no tasks: no masks)
  const long long blocks = std::min<long long>(((long long)g->nv * 8 + 255) / 256, (long long)g->cu_count * 32);
  hipLaunchKernelGGL(sup_mask_size_kernel, dim3((unsigned)std::max<long long>(1, blocks)), dim3(256), 0, 0, tw, sup_mask_min_tail(g), off.p);
  HIP_TRY(dev_exclusive_sum(tmp, off.p, off.p, ne + 1));
  unsigned long long total = 0;
  HIP_TRY(hipMemcpy(&total, off.p + ne, sizeof total, hipMemcpyDeviceToHost));
  setup_trace("support masks: sizes + scan");
  if (total == 0 || total >= (unsigned long long)kNoMask) {
    g->smask_state = 2;
    g->setup.table_ms += timer.ms();
    return GM_OK;
  }
  // the rows that have a second mask word, from the top of the id range
  DevBuf<int> fflag, fpos;
  HIP_TRY(fflag.alloc((size_t)g->nv + 1));
  HIP_TRY(fpos.alloc((size_t)g->nv + 1));
  HIP_TRY(hipMemsetAsync(fflag.p, 0, sizeof(int) * ((size_t)g->nv + 1), 0));
  hipLaunchKernelGGL(sup_far_flag_kernel, dim3((unsigned)((g->nv + 255) / 256)), dim3(256), 0, 0, g->nv, g->d_rp, sup_mask_min_tail(g), kTctStageMax, g->tl_skip_from, fflag.p);
  HIP_TRY(dev_exclusive_sum(tmp, fflag.p, fpos.p, (size_t)g->nv + 1));
  int nfar = 0;
  HIP_TRY(hipMemcpy(&nfar, fpos.p + g->nv, sizeof nfar, hipMemcpyDeviceToHost));
  unsigned *emoff = nullptr, *tmoff = nullptr;
  unsigned long long *arena = nullptr;
  int *far_rows = nullptr;
  hipError_t e = dev_malloc(&emoff, sizeof(unsigned) * ne);
  if (e == hipSuccess) e = dev_malloc(&far_rows, sizeof(int) * (size_t)std::max(nfar, 1));
  if (e == hipSuccess) hipLaunchKernelGGL(sup_far_list_kernel, dim3((unsigned)((g->nv + 255) / 256)), dim3(256), 0, 0, g->nv, fflag.p, fpos.p, far_rows);
  if (e == hipSuccess) e = dev_malloc(&tmoff, sizeof(unsigned) * ne);
  if (e == hipSuccess) e = dev_malloc(&arena, sizeof(unsigned long long) * (size_t)total);
  // (every launch rewrites the words of the tasks it runs; a launch that leaves masked tasks out -- a chunk filter, an aborted launch --
  // must find zeros, not what hipMalloc handed over: ADVICE r5)
  if (e == hipSuccess) e = hipMemsetAsync(arena, 0, sizeof(unsigned long long) * (size_t)total, 0);
  if (e == hipSuccess) {
    const long long eb = std::min<long long>(((long long)ne + 255) / 256, (long long)g->cu_count * 32);
    hipLaunchKernelGGL(sup_mask_off_kernel, dim3((unsigned)eb), dim3(256), 0, 0, (long long)ne, off.p, emoff);
    hipLaunchKernelGGL(sup_mask_task_kernel, dim3((unsigned)eb), dim3(256), 0, 0, (long long)ne, g->d_tdesc, g->d_tedge, emoff, tmoff);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {  // (an optimisation that did not fit: the atomics stay)
    for (void *q : {(void *)emoff, (void *)tmoff, (void *)arena, (void *)far_rows})
      if (q) dev_free(q);
    (void)hipGetLastError();
    g->smask_state = 2;
    g->setup.table_ms += timer.ms();
    return GM_OK;
  }
  g->d_emoff = emoff;
  g->d_tmoff = tmoff;
  g->d_smask = arena;
  g->d_sup_far_rows = far_rows;
  g->n_sup_far_rows = nfar;
  g->smask_words = total;
  g->smask_state = 1;
  setup_trace("support masks: offsets");
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// the rows beyond the stage of the task-list kernels: a handful per graph, found on the host copy of the offsets
int ensure_long_rows(gm_graph *g) {
  if (g->n_long_rows >= 0) return GM_OK;
  const std::vector<int> *rp = nullptr;
  int rc = host_rp(g, &rp);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->n_long_rows >= 0) return GM_OK;
  std::vector<int> rows;
  std::vector<long long> prefix(1, 0);
  for (int v = 0; v < g->nv; ++v) {
    const int d = (*rp)[(size_t)v + 1] - (*rp)[(size_t)v];
    if (d > kTctStageMax) {
      rows.push_back(v);
      prefix.push_back(prefix.back() + d);
    }
  }
  if (!rows.empty()) {
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(dev_malloc(&g->d_long_rows, sizeof(int) * rows.size()));
    HIP_TRY(dev_malloc(&g->d_long_prefix, sizeof(long long) * prefix.size()));
    HIP_TRY(copy_to_device(g->d_long_rows, rows.data(), sizeof(int) * rows.size()));
    HIP_TRY(copy_to_device(g->d_long_prefix, prefix.data(), sizeof(long long) * prefix.size()));
  }
  g->long_edges = prefix.back();
  g->n_long_rows = (int)rows.size();
  return GM_OK;
}

int ensure_edesc(gm_graph *g) {
  if (g->d_edesc || g->ne == 0) return GM_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->d_edesc) return GM_OK;
  SetupTimer timer;
  int2 *d = nullptr;
  HIP_TRY(dev_malloc(&d, sizeof(int2) * (size_t)g->ne));
  const long long blocks = std::min<long long>((g->ne + 255) / 256, (long long)g->cu_count * 32);
  hipLaunchKernelGGL(edesc_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->ne, g->d_rp, g->d_col, d);
  hipError_t e = hipDeviceSynchronize();
  setup_trace("edge descriptors");
  if (e != hipSuccess) { dev_free(d); return hip_fail(e, "edesc_kernel", __FILE__, __LINE__); }
  g->d_edesc = d;
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// k-clique (k = 4): the plan of one rank's share for the re-hosted first level (gm_mine.h, gm_cbuild.hip). Everything below runs on
// the device except what is O(wide vertices of the share) or O(chunks).
#ifndef GM_WIDE_ARENA_MB
#define GM_WIDE_ARENA_MB 16384
#endif
__global__ __launch_bounds__(256) void wide_flag_kernel(int nv, const int *__restrict__ rp, int min_words, int *__restrict__ flag, int *__restrict__ iota) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  flag[v] = clique_is_wide(rp[v + 1] - rp[v], min_words) ? 1 : 0;
  iota[v] = v;
}
// this rank's share of the sorted wide list: slot i = entry first + i * step
__global__ __launch_bounds__(256) void wide_share_kernel(int count, long long first, long long step, const int *__restrict__ wide_sorted,
                                                         const int *__restrict__ rp, int *__restrict__ verts, int *__restrict__ degs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int u = wide_sorted[first + (long long)i * step];
  verts[i] = u;
  degs[i] = rp[u + 1] - rp[u];
}
__device__ __forceinline__ unsigned long long cb_matrix_words(int d) {
  return (d >= kCbMinDeg && d <= kCbMaxDeg) ? (unsigned long long)d * (unsigned long long)((d + 31) / 32) : 0ull;
}
// words of the matrices of every narrow chunk of the share (position i of the share = chunk order[first + i * step])
__global__ __launch_bounds__(256) void cb_chunk_words_kernel(long long count, long long first, long long step, const int *__restrict__ order,
                                                             const ChunkRec *__restrict__ chunks, const int *__restrict__ rp,
                                                             unsigned long long *__restrict__ words) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const long long pos = first + i * step;
  const ChunkRec r = chunks[order ? order[pos] : pos];
  unsigned long long w = 0;
  for (int u = r.u_begin; u < r.u_end; ++u) w += cb_matrix_words(rp[u + 1] - rp[u]);
  words[i] = w;
}
// owners of a round: the vertices of its narrow chunks ...
__global__ __launch_bounds__(256) void cb_own_chunks_kernel(long long count, long long first, long long step, const int *__restrict__ order,
                                                            const ChunkRec *__restrict__ chunks, int *__restrict__ own) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const long long pos = first + i * step;
  const ChunkRec r = chunks[order ? order[pos] : pos];
  for (int u = r.u_begin; u < r.u_end; ++u) own[u] = 1;
}
// ... and its wide vertices
__global__ __launch_bounds__(256) void cb_own_verts_kernel(int count, const int *__restrict__ verts, int *__restrict__ own) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) own[verts[i]] = 2;  // (2: a wide vertex -- its rows with a first endpoint in the core are gathered, cb_owner_sizes_kernel)
}
// per vertex: matrix words and task edges of an owner (0 for everybody else; [nv] = 0 so that the scans end with the totals)
// core_base >= 0: a wide owner's rows whose first endpoint is >= core_base come from the core bitmap (gm_cgather.hip): only the
// entries below it -- the first ones of the ascending row -- are tasks of the streamed build
__global__ __launch_bounds__(256) void cb_owner_sizes_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ col, const int *__restrict__ own,
                                                             int core_base, unsigned long long *__restrict__ words, int *__restrict__ tasks) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nv) return;
  unsigned long long w = 0;
  int t = 0;
  if (v < nv && own[v]) {
    const int d = rp[v + 1] - rp[v];
    w = cb_matrix_words(d);
    t = w ? d : 0;
    if (t && own[v] == 2 && core_base >= 0) t = lower_bound(col + rp[v], d, core_base);
  }
  words[v] = w;
  tasks[v] = t;
}
// The task records of a round by counting placement (like ensure_tasklists: until round 4 one (host, entry) key per task found by a
// bisection per entry, a radix sort of 109 M key / owner pairs on the com-Orkut stand-in, a record pass).  Eight lanes walk the row of an
// owner u: the first ntask_of[u] entries are its streamed tasks (all of them, or -- a wide vertex beside a core bitmap -- those below the
// core).  The edge u -> v is hosted by the endpoint whose list is NOT streamed: N+(v) whole, or N+(u) -- beyond v under a topological
// numbering -- whichever is shorter (round 3 compared the whole lists: 15 % more keys on R-MAT), when it fits the stage.
// (same-address atomics aggregated like task_rows_kernel's: per 8-lane group for the tasks a row hosts itself, per workgroup in an LDS
// histogram for the in-edge tasks of the last kHubWin hosts of a topologically numbered DAG)
template <bool PLACE>
__global__ __launch_bounds__(256) void cb_task_rows_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ col, const int2 *__restrict__ edesc,
                                                           const int *__restrict__ ntask_of, int topo, int hub0, int *__restrict__ cnt /* PLACE: the cursors */,
                                                           const int *__restrict__ trp, const unsigned long long *__restrict__ base,
                                                           CBuildTask *__restrict__ out) {
  __shared__ int hist[kHubWin];
  for (int h = threadIdx.x; h < kHubWin; h += 256) hist[h] = 0;
  __syncthreads();
  const long long stride = ((long long)gridDim.x * blockDim.x) >> 3;
  const int lane = threadIdx.x & 63, sub = lane & 7, g8 = lane & ~7;
  const long long u0 = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) >> 3;
  auto put = [&](const int slot, const int u, const int i, const int ru, const int du, const int2 dv, const bool v_hosts) {
    const int words = (du + 31) / 32;
    const unsigned long long off = base[u] + (unsigned long long)i * (unsigned long long)words;
    CBuildTask T;
    unsigned fl = (unsigned)(off >> 32) & 255u;
    if (!v_hosts) {  // type A: N+(v) is streamed against the staged N+(u)
      T.list = dv.x;
      T.len = dv.y;
    } else {         // type B: N+(u) -- beyond v when the numbering is topological -- is streamed against the staged N+(v)
      const int skip = topo ? i + 1 : 0;
      T.list = ru + skip;
      T.len = du - skip;
      fl |= ((unsigned)skip << 8) | 0x80000000u;
    }
    T.off_lo = (unsigned)off;
    T.off_hi_fl = fl | ((unsigned)words << 20);
    out[slot] = T;
  };
  for (int phase = 0; phase < (PLACE ? 2 : 1); ++phase) {
    for (long long ub = u0; ub < nv; ub += stride) {
      const long long u = ub + (lane >> 3);
      int nt = 0, ru = 0, du = 0;
      if (u < nv) {
        nt = ntask_of[u];  // 0: not an owner of this round, or no matrix (d+ < 3 / > kCbMaxDeg), or every row comes from the core bitmap
        ru = rp[u];
        du = rp[u + 1] - ru;
      }
      const int nmax = wave_max_nonneg(nt);
      for (int i = sub; i - sub < nmax; i += 8) {  // wave-uniform trip count
        const bool act = i < nt;
        const int e = ru + (act ? i : 0);
        const int2 dv = act ? edesc[e] : make_int2(0, 0);  // {rp[v], d+(v)}
        const int tail = topo ? du - i - 1 : du;
        const bool v_hosts = dv.y > tail && dv.y <= kCbMaxDeg;
        if (phase == 0) {
          const unsigned long long mu = __ballot(act && !v_hosts);
          const unsigned gm = (unsigned)(mu >> g8) & 255u;
          int ubase = 0;
          if (gm != 0u && sub == (int)__builtin_ctz(gm)) ubase = atomicAdd(&cnt[u], (int)__builtin_popcount(gm));
          if (PLACE && gm != 0u) {
            ubase = __shfl(ubase, g8 + (int)__builtin_ctz(gm));
            if (act && !v_hosts) put(trp[u] + ubase + (int)__builtin_popcount(gm & ((1u << sub) - 1u)), (int)u, i, ru, du, dv, false);
          }
        }
        if (act && v_hosts) {
          const int host = col[e];
          if (host >= hub0) {
            if (phase == 0) atomicAdd(&hist[host - hub0], 1);
            else put(trp[host] + atomicAdd(&hist[host - hub0], 1), (int)u, i, ru, du, dv, true);
          } else if (phase == 0) {
            if (!PLACE) atomicAdd(&cnt[host], 1);
            else put(trp[host] + atomicAdd(&cnt[host], 1), (int)u, i, ru, du, dv, true);
          }
        }
      }
    }
    if (phase == 0) {
      __syncthreads();
      for (int h = threadIdx.x; h < kHubWin; h += 256) {
        const int c = hist[h];
        if (c) {
          const int b = atomicAdd(&cnt[hub0 + h], c);  // pass 1: the count; pass 2: this workgroup's range among the hub's tasks
          if (PLACE) hist[h] = b;  // (the ranks inside the range are taken from the same word)
        }
      }
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(256) void cb_slot_base_kernel(int w0, int w1, const int *__restrict__ verts, const unsigned long long *__restrict__ base,
                                                           unsigned long long *__restrict__ slot_base) {
  const int s = w0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (s < w1) slot_base[s] = base[verts[s]];
}

// ---- core bitmap: dense adjacency of the last core_h vertices of a topologically numbered DAG (gm_host.h; gathered by gm_cgather.hip) ----
__global__ __launch_bounds__(256) void core_fill_kernel(int nv, int base, int words, long long e0, long long e1, const int *__restrict__ rp,
                                                         const int *__restrict__ col, unsigned *__restrict__ bits) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += stride) {
    int lo = base, hi = nv - 1;  // the row of entry e
    while (lo < hi) {
      const int mid = (int)(((long long)lo + hi + 1) >> 1);
      if (rp[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int w = col[e] - base;  // (topological: w > lo - base >= 0)
    atomicOr(&bits[(size_t)(lo - base) * (size_t)words + (size_t)(w >> 5)], 1u << (w & 31));
  }
}
int ensure_core_bitmap(gm_graph *g) {
  if (g->core_state) return GM_OK;
  bool topo = false;
  int rc = graph_is_topological(g, &topo);
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->core_state) return GM_OK;
  long long want = kCoreHDefault;
  if (const char *e = gm_sweep_env("GM_CORE_H")) want = atoll(e);  // (sweeps; 0 switches the gathered build off)
  if (!topo || g->d_rp == nullptr || g->nv < 64 || want < 64) {
    g->core_state = 2;
    return GM_OK;
  }
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const int h = (int)std::min<long long>((long long)g->nv, want);
  const int base = g->nv - h;
  const int words = (h + 31) / 32;
  const size_t bytes = (size_t)h * (size_t)words * 4;
  if (dev_malloc(&g->d_core, bytes) != hipSuccess) {  // no room: the streamed build does everything
    (void)hipGetLastError();
    g->d_core = nullptr;
    g->core_state = 2;
    return GM_OK;
  }
  HIP_TRY(hipMemsetAsync(g->d_core, 0, bytes, 0));
  int e01[2] = {0, 0};
  HIP_TRY(hipMemcpy(&e01[0], g->d_rp + base, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&e01[1], g->d_rp + g->nv, sizeof(int), hipMemcpyDeviceToHost));
  const long long n = (long long)e01[1] - e01[0];
  if (n > 0) {
    const long long blocks = std::min<long long>((n + 255) / 256, (long long)g->cu_count * 32);
    hipLaunchKernelGGL(core_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->nv, base, words, (long long)e01[0], (long long)e01[1], g->d_rp, g->d_col, g->d_core);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  g->core_h = h;
  g->core_base = base;
  setup_trace("clique: core bitmap");
  g->core_state = 1;
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// ---- the core bitmap cut into blocks of rows for the blocked gather (gm_mine.h CGatherBParams; gm_cgather.hip cgatherb_kernel) ----------
// A block = consecutive core rows whose stored words (row p keeps the words from cgb_first_word(p) on: the columns right of its diagonal)
// fit kCgbWords, at most kCgbMaxRows of them: 16 rows of 4 KB at the bottom of a 32 K core, ~1000 rows of a few words at its top -- where
// the rows that most vertices hold are.
// (bit positions are taken relative to the core's base rounded DOWN to a multiple of 32 -- `delta` = core_base & 31 phantom columns in front --
// so that the word of a column is (vertex id >> 5) minus a constant whatever the graph's size: the kernel folds the constant into the rows'
// LDS offsets and never subtracts the base from a column)
__global__ __launch_bounds__(256) void core_tri_fill_kernel(int nv, int base, long long e0, long long e1, const int *__restrict__ rp, const int *__restrict__ col,
                                                             const int *__restrict__ bid, const int *__restrict__ rowbase, const int4 *__restrict__ blk,
                                                             unsigned *__restrict__ tri) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int base32 = base & ~31;
  for (long long e = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < e1; e += stride) {
    int lo = base, hi = nv - 1;  // the row of entry e
    while (lo < hi) {
      const int mid = (int)(((long long)lo + hi + 1) >> 1);
      if (rp[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int p = lo - base, q = col[e] - base32;  // (topological: col[e] > lo)
    atomicOr(&tri[(long long)blk[bid[p]].z + rowbase[p] + (q >> 5)], 1u << (q & 31));
  }
}
int ensure_core_tri(gm_graph *g) {
  if (g->cg_tri_state) return GM_OK;
  {
    const int rc = ensure_core_bitmap(g);  // (takes the lock itself)
    if (rc) return rc;
  }
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->cg_tri_state) return GM_OK;
  if (g->core_state != 1 || g->core_h < 64) {
    g->cg_tri_state = 2;
    return GM_OK;
  }
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const int h = g->core_h, delta = g->core_base & 31, words = cgb_words(h, delta);
  std::vector<int> bid((size_t)h), rowbase((size_t)h);
  std::vector<int4> blk;
  long long total = 0;
  {
    int p0 = 0, used = 0;
    for (int p = 0; p <= h; ++p) {
      const int span = p < h ? words - cgb_first_word(p, delta) : 0;
      if (p == h || used + span > kCgbWords || p - p0 >= kCgbMaxRows) {  // close the block [p0, p)
        if (p > p0) {
          blk.push_back(make_int4(p0, p - p0, (int)total, used));
          total += (used + 3) & ~3;  // (images start at multiples of four words: 16-byte copies)
        }
        p0 = p;
        used = 0;
        if (p == h) break;
      }
      bid[(size_t)p] = (int)blk.size();
      rowbase[(size_t)p] = used - cgb_first_word(p, delta);
      used += span;
    }
  }
  if (total >= (1ll << 31) || blk.size() >= 65536) {
    g->cg_tri_state = 2;
    return GM_OK;
  }
  unsigned *tri = nullptr;
  int *d_bid = nullptr, *d_rowbase = nullptr;
  int4 *d_blk = nullptr;
  int e01[2] = {0, 0};
  hipError_t e = dev_malloc(&tri, sizeof(unsigned) * (size_t)std::max<long long>(total, 4));
  if (e == hipSuccess) e = dev_malloc(&d_bid, sizeof(int) * (size_t)h);
  if (e == hipSuccess) e = dev_malloc(&d_rowbase, sizeof(int) * (size_t)h);
  if (e == hipSuccess) e = dev_malloc(&d_blk, sizeof(int4) * blk.size());
  if (e == hipSuccess) e = hipMemsetAsync(tri, 0, sizeof(unsigned) * (size_t)std::max<long long>(total, 4), 0);
  if (e == hipSuccess) e = copy_to_device(d_bid, bid.data(), sizeof(int) * (size_t)h);
  if (e == hipSuccess) e = copy_to_device(d_rowbase, rowbase.data(), sizeof(int) * (size_t)h);
  if (e == hipSuccess) e = copy_to_device(d_blk, blk.data(), sizeof(int4) * blk.size());
  if (e == hipSuccess) e = hipMemcpy(&e01[0], g->d_rp + g->core_base, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&e01[1], g->d_rp + g->nv, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) {
    const long long n = (long long)e01[1] - e01[0];
    if (n > 0)
      hipLaunchKernelGGL(core_tri_fill_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, (long long)g->cu_count * 32)), dim3(256), 0, 0, g->nv, g->core_base,
                         (long long)e01[0], (long long)e01[1], g->d_rp, g->d_col, d_bid, d_rowbase, d_blk, tri);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {  // (no room: the row-major gather does the work)
    for (void *q : {(void *)tri, (void *)d_bid, (void *)d_rowbase, (void *)d_blk})
      if (q) dev_free(q);
    (void)hipGetLastError();
    g->cg_tri_state = 2;
    return GM_OK;
  }
  g->d_cg_tri = tri;
  g->d_cg_bid = d_bid;
  g->d_cg_rowbase = d_rowbase;
  g->d_cg_blk = d_blk;
  g->n_cg_blocks = (int)blk.size();
  g->cg_tri_state = 1;
  setup_trace("clique: core blocks");
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

__global__ __launch_bounds__(256) void iota_u32_kernel(long long n, unsigned *__restrict__ out) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = (unsigned)k;
}
// The units of a round's wide slots.  One wave per slot walks the core part of the vertex's row: a unit starts where the block of the
// row's core position changes.  EMIT = false: units per slot;  EMIT = true: the records {start of the vertex's column table, matrix offset, d | first row << 16, rows}
// + the block as sort key, at the slot's offset.
template <bool EMIT>
__global__ __launch_bounds__(256) void cgb_units_kernel(int w0, int nslots, const int *__restrict__ verts, const int *__restrict__ rp, const int *__restrict__ col,
                                                         const unsigned long long *__restrict__ slot_base, int core_base, const int *__restrict__ bid,
                                                         int *__restrict__ cnt, const int *__restrict__ uoff, const unsigned *__restrict__ poff,
                                                         unsigned *__restrict__ keys, uint4 *__restrict__ recs) {
  const int lane = threadIdx.x & 63;
  const int s = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (s >= nslots) return;  // (wave-uniform)
  const int u = verts[w0 + s];
  const int ru = rp[u], d = rp[u + 1] - ru;
  const int k0 = lower_bound(col + ru, d, core_base);
  int n = 0;
  const int out0 = EMIT ? uoff[s] : 0;
  const unsigned mbase = EMIT ? (unsigned)slot_base[w0 + s] : 0u;
  for (int jb = k0; jb < d; jb += 64) {
    const int j = jb + lane;
    const int b = j < d ? bid[col[ru + j] - core_base] : -1;
    const int bp = (j > k0 && j < d) ? bid[col[ru + j - 1] - core_base] : -2;
    const bool start = j < d && b != bp;
    const unsigned long long m = __ballot(start);
    if (EMIT && start) {
      // rows of the unit: up to the next start in this tile, else scan ahead (units are short: a few rows)
      int e = j + 1;
      while (e < d && bid[col[ru + e] - core_base] == b) ++e;
      const int k = out0 + n + (int)__popcll(m & ((1ull << lane) - 1ull));
      keys[k] = (unsigned)b;
      recs[k] = make_uint4(poff[s], mbase, (unsigned)d | ((unsigned)j << 16), (unsigned)(e - j));
    }
    n += (int)__popcll(m);
  }
  if (!EMIT && lane == 0) cnt[s] = n;
}
// The COLUMN TABLES the blocked gather reads instead of col: per wide slot, the positions of N+(u) as 16-bit numbers q = id - (core_base & ~31)
// (0: a neighbour below the core, or padding), swizzled so that ONE dword load per lane fetches two column tiles -- dword 64 T + l of a vertex
// = q(128 T + l) | q(128 T + 64 + l) << 16: half the bytes of col and half the loads.
__global__ __launch_bounds__(256) void cgb_table_size_kernel(int w0, int nslots, const int *__restrict__ verts, const int *__restrict__ rp, unsigned *__restrict__ sz) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nslots) return;
  unsigned v = 0u;
  if (s < nslots) {
    const int u = verts[w0 + s];
    const int d = rp[u + 1] - rp[u];
    v = 64u * (unsigned)(((d + 63) / 64 + 1) / 2);
  }
  sz[s] = v;
}
__global__ __launch_bounds__(256) void cgb_table_fill_kernel(int w0, int nslots, const int *__restrict__ verts, const int *__restrict__ rp, const int *__restrict__ col,
                                                              int core_base, const unsigned *__restrict__ poff, unsigned *__restrict__ tab) {
  const int lane = threadIdx.x & 63;
  const int s = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (s >= nslots) return;  // (wave-uniform)
  const int u = verts[w0 + s];
  const int ru = rp[u], d = rp[u + 1] - ru, base32 = core_base & ~31;
  unsigned *__restrict__ out = tab + poff[s];
  const int npairs = ((d + 63) / 64 + 1) / 2;
  for (int t = 0; t < npairs; ++t) {
    const int j0 = 128 * t + lane, j1 = j0 + 64;
    const int c0 = j0 < d ? col[ru + j0] : 0, c1 = j1 < d ? col[ru + j1] : 0;
    const unsigned q0 = c0 >= core_base ? (unsigned)(c0 - base32) : 0u, q1 = c1 >= core_base ? (unsigned)(c1 - base32) : 0u;
    out[64 * t + lane] = q0 | (q1 << 16);
  }
}
// sorted order: record k = recs[idx[k]]; its probes = sum over its rows i of (d - 1 - i)
__global__ __launch_bounds__(256) void cgb_gather_kernel(long long n, const unsigned *__restrict__ idx, const uint4 *__restrict__ recs, uint4 *__restrict__ out,
                                                          unsigned long long *__restrict__ cost, unsigned long long *__restrict__ tbytes) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint4 r = recs[idx[k]];
  out[k] = r;
  const long long d = r.z & 0xffffu, i0 = r.z >> 16, rows = r.w;
  cost[k] = (unsigned long long)(rows * (d - 1 - i0) - rows * (rows - 1) / 2 + 64 * rows + 256);  // (+ per row and per unit overheads, in probes)
  // what the unit reads by construction: its record, its rows' positions, the table dwords of the tile pairs from its first row's on
  const long long npairs = ((d + 63) / 64 + 1) / 2, t0 = (i0 + 1) >> 7;
  tbytes[k] = (unsigned long long)(16 + 4 * rows + 256 * (npairs > t0 ? npairs - t0 : 0));
}
__global__ __launch_bounds__(256) void cgb_item_flag_kernel(long long n, const unsigned *__restrict__ keys, const unsigned long long *__restrict__ cum,
                                                             unsigned long long target, int *__restrict__ flag) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  flag[k] = (k == 0 || keys[k] != keys[k - 1] || cum[k] / target != cum[k - 1] / target) ? 1 : 0;
}
__global__ __launch_bounds__(256) void cgb_item_place_kernel(long long n, const unsigned *__restrict__ keys, const int *__restrict__ flag, const int *__restrict__ pos,
                                                              int2 *__restrict__ items, int n_items) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) items[n_items] = make_int2((int)n, -1);
  if (k >= n) return;
  if (flag[k]) items[pos[k]] = make_int2((int)k, (int)keys[k]);
}
#ifndef GM_CGB_ITEM_COST
#define GM_CGB_ITEM_COST (1ull << 20)
#endif
constexpr unsigned long long kCgbItemCost = GM_CGB_ITEM_COST;  // probes (+ overheads) of a work item: ~80 us of one workgroup
static int build_gather_index(gm_graph *g, CliquePlan &pl, CliqueRound &rd, ScanTemp &tmp) {
  const int nslots = (int)(rd.w1 - rd.w0);
  if (nslots <= 0 || pl.core_base < 0 || g->cg_tri_state != 1 || rd.words >= (1ull << 32)) return GM_OK;  // (no index: the row-major gather)
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  DevBuf<int> cnt, uoff;
  HIP_TRY(cnt.alloc((size_t)nslots + 1));
  HIP_TRY(uoff.alloc((size_t)nslots + 1));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(int) * ((size_t)nslots + 1), 0));
  const dim3 wgrid((unsigned)(((long long)nslots + 3) / 4));
  hipLaunchKernelGGL((cgb_units_kernel<false>), wgrid, dim3(256), 0, 0, (int)rd.w0, nslots, pl.d_verts, g->d_rp, g->d_col, pl.d_slot_base, pl.core_base, g->d_cg_bid,
                     cnt.p, nullptr, nullptr, nullptr, nullptr);
  HIP_TRY(dev_exclusive_sum(tmp, cnt.p, uoff.p, (size_t)nslots + 1));
  int nu = 0;
  HIP_TRY(hipMemcpy(&nu, uoff.p + nslots, sizeof(int), hipMemcpyDeviceToHost));
  if (nu <= 0) return GM_OK;
  // the column tables of the round's slots
  DevBuf<unsigned> tsz, poff;
  HIP_TRY(tsz.alloc((size_t)nslots + 1));
  HIP_TRY(poff.alloc((size_t)nslots + 1));
  hipLaunchKernelGGL(cgb_table_size_kernel, blocks((long long)nslots + 1), dim3(256), 0, 0, (int)rd.w0, nslots, pl.d_verts, g->d_rp, tsz.p);
  HIP_TRY(dev_exclusive_sum(tmp, tsz.p, poff.p, (size_t)nslots + 1));
  unsigned tab_words = 0;
  HIP_TRY(hipMemcpy(&tab_words, poff.p + nslots, sizeof(unsigned), hipMemcpyDeviceToHost));
  unsigned *tab = nullptr;
  // (+ nine tile pairs of slack: the kernel requests nine pairs from a unit's first one on without looking at the table's end -- what lies
  // beyond belongs to tiles the vertex does not have and is never used; byte offsets are 32 bits)
  if ((unsigned long long)tab_words * 4ull + 4096ull >= (1ull << 32)) return GM_OK;  // (no index: the row-major gather)
  HIP_TRY(dev_malloc(&tab, sizeof(unsigned) * ((size_t)std::max(tab_words, 64u) + 9 * 64)));
  hipLaunchKernelGGL(cgb_table_fill_kernel, wgrid, dim3(256), 0, 0, (int)rd.w0, nslots, pl.d_verts, g->d_rp, g->d_col, pl.core_base, poff.p, tab);
  struct TabGuard {  // (the table belongs to the round once the index is complete)
    unsigned *p;
    ~TabGuard() { if (p) dev_free(p); }
  } tab_guard{tab};
  DevBuf<unsigned> keys, keys_s, idx, idx_s;
  DevBuf<uint4> recs;
  DevBuf<unsigned long long> cost, cum;
  DevBuf<int> flag, pos;
  HIP_TRY(keys.alloc((size_t)nu));
  HIP_TRY(keys_s.alloc((size_t)nu));
  HIP_TRY(idx.alloc((size_t)nu));
  HIP_TRY(idx_s.alloc((size_t)nu));
  HIP_TRY(recs.alloc((size_t)nu));
  hipLaunchKernelGGL((cgb_units_kernel<true>), wgrid, dim3(256), 0, 0, (int)rd.w0, nslots, pl.d_verts, g->d_rp, g->d_col, pl.d_slot_base, pl.core_base, g->d_cg_bid,
                     nullptr, uoff.p, poff.p, keys.p, recs.p);
  hipLaunchKernelGGL(iota_u32_kernel, blocks(nu), dim3(256), 0, 0, (long long)nu, idx.p);
  {
    int bits = 1;
    while ((1 << bits) < g->n_cg_blocks) ++bits;
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys.p, keys_s.p, idx.p, idx_s.p, nu, 0, bits));
    HIP_TRY(tmp.reserve(bytes));
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.buf.p, bytes, keys.p, keys_s.p, idx.p, idx_s.p, nu, 0, bits));  // (stable: a block's units stay in slot order)
  }
  uint4 *units = nullptr;
  HIP_TRY(dev_malloc(&units, sizeof(uint4) * (size_t)nu));
  hipError_t e = cost.alloc((size_t)nu + 1);
  if (e == hipSuccess) e = cum.alloc((size_t)nu + 1);
  if (e == hipSuccess) e = flag.alloc((size_t)nu + 1);
  if (e == hipSuccess) e = pos.alloc((size_t)nu + 1);
  int ni = 0;
  int2 *items = nullptr;
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cgb_gather_kernel, blocks(nu), dim3(256), 0, 0, (long long)nu, idx_s.p, recs.p, units, cost.p, cum.p);
    size_t bytes = 0;  // (cum holds the units' table bytes for a moment: their sum is the kernel's own bytes, gm_clique4_gather_info)
    DevBuf<unsigned long long> tsum;
    e = tsum.alloc(1);
    if (e == hipSuccess) e = hipcub::DeviceReduce::Sum(nullptr, bytes, cum.p, tsum.p, nu);
    if (e == hipSuccess) e = tmp.reserve(bytes);
    if (e == hipSuccess) e = hipcub::DeviceReduce::Sum(tmp.buf.p, bytes, cum.p, tsum.p, nu);
    if (e == hipSuccess) e = hipMemcpy(&rd.gather_table_bytes, tsum.p, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = dev_exclusive_sum(tmp, cost.p, cum.p, (size_t)nu);
  }
  if (e == hipSuccess) e = hipMemsetAsync(flag.p + nu, 0, sizeof(int), 0);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cgb_item_flag_kernel, blocks(nu), dim3(256), 0, 0, (long long)nu, keys_s.p, cum.p, kCgbItemCost, flag.p);
    e = dev_exclusive_sum(tmp, flag.p, pos.p, (size_t)nu + 1);
  }
  if (e == hipSuccess) e = hipMemcpy(&ni, pos.p + nu, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = dev_malloc(&items, sizeof(int2) * ((size_t)ni + 1));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(cgb_item_place_kernel, blocks(nu), dim3(256), 0, 0, (long long)nu, keys_s.p, flag.p, pos.p, items, ni);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) {
    dev_free(units);
    if (items) dev_free(items);
    return hip_fail(e, "gather index", __FILE__, __LINE__);
  }
  rd.d_gunits = units;
  rd.d_gtab = tab;
  tab_guard.p = nullptr;
  rd.d_gitems = items;
  rd.n_gunits = (size_t)nu;
  rd.n_gitems = (size_t)ni;
  return GM_OK;
}

int clique_wide_min_words() {
  static const int v = [] {
    const char *e = gm_sweep_env("GM_WIDE_MIN_WORDS");  // (sweeps; read once: tables and plans are cached per graph)
    return e ? std::max(64, std::min(atoi(e), kBitWords)) : kWideMinWordsDefault;
  }();
  return v;
}

__global__ __launch_bounds__(256) void mcls_rec_kernel(int n, const int *__restrict__ slots, const int *__restrict__ verts, const int *__restrict__ rp,
                                                       const unsigned long long *__restrict__ slot_base, int4 *__restrict__ rec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int slot = slots[k], u = verts[slot];
  const unsigned long long b = slot_base[slot];
  rec[k] = make_int4(rp[u + 1] - rp[u], (int)(unsigned)b, (int)(unsigned)(b >> 32), 0);
}
static void free_clique_plan(CliquePlan &pl) {
  if (pl.d_verts) dev_free(pl.d_verts);
  if (pl.d_slot_base) dev_free(pl.d_slot_base);
  if (pl.d_mcls_slots) dev_free(pl.d_mcls_slots);
  if (pl.d_mcls_rec) dev_free(pl.d_mcls_rec);
  for (auto &rd : pl.rounds) {
    if (rd.d_base) dev_free(rd.d_base);
    if (rd.d_tasks) dev_free(rd.d_tasks);
    if (rd.d_trp) dev_free(rd.d_trp);
    if (rd.d_gunits) dev_free(rd.d_gunits);
    if (rd.d_gtab) dev_free(rd.d_gtab);
    if (rd.d_gitems) dev_free(rd.d_gitems);
    free_table(rd.host_tab);
  }
  pl.rounds.clear();
  pl.d_verts = nullptr; pl.d_slot_base = nullptr; pl.d_mcls_slots = nullptr; pl.d_mcls_rec = nullptr;
}
void free_clique_plans(gm_graph *g) {
  for (auto &pl : g->clique_plans) free_clique_plan(pl);
  g->clique_plans.clear();
}

// one round: owners -> offsets -> task lists -> host-chunk table
static int build_clique_round(gm_graph *g, CliquePlan &pl, CliqueRound &rd, ScanTemp &tmp) {
  const int nv = g->nv;
  const size_t nv1 = (size_t)nv + 1;
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  DevBuf<int> own, ntask_of, tpos, cnt;
  DevBuf<unsigned long long> words;
  HIP_TRY(own.alloc(nv1));
  HIP_TRY(ntask_of.alloc(nv1));
  HIP_TRY(tpos.alloc(nv1));
  HIP_TRY(words.alloc(nv1));
  HIP_TRY(hipMemsetAsync(own.p, 0, sizeof(int) * nv1, 0));
  if (rd.n_count > 0)
    hipLaunchKernelGGL(cb_own_chunks_kernel, blocks(rd.n_count), dim3(256), 0, 0, rd.n_count, pl.n_first + rd.n_pos0 * pl.n_step, pl.n_step, pl.d_order, pl.tabN->d, own.p);
  if (rd.w1 > rd.w0)
    hipLaunchKernelGGL(cb_own_verts_kernel, blocks((long long)(rd.w1 - rd.w0)), dim3(256), 0, 0, (int)(rd.w1 - rd.w0), pl.d_verts + rd.w0, own.p);
  hipLaunchKernelGGL(cb_owner_sizes_kernel, blocks((long long)nv1), dim3(256), 0, 0, nv, g->d_rp, g->d_col, own.p, pl.core_base, words.p, ntask_of.p);
  HIP_TRY(dev_malloc(&rd.d_base, sizeof(unsigned long long) * nv1));
  HIP_TRY(dev_exclusive_sum(tmp, words.p, rd.d_base, nv1));
  HIP_TRY(dev_exclusive_sum(tmp, ntask_of.p, tpos.p, nv1));
  int nt = 0;
  HIP_TRY(hipMemcpy(&rd.words, rd.d_base + nv, 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&nt, tpos.p + nv, sizeof(int), hipMemcpyDeviceToHost));
  rd.n_tasks = (size_t)nt;
  HIP_TRY(dev_malloc(&rd.d_trp, sizeof(int) * nv1));
  if (rd.w1 > rd.w0)  // (before the early return: a round whose wide rows all come from the core bitmap has no streamed task at all)
    hipLaunchKernelGGL(cb_slot_base_kernel, blocks((long long)(rd.w1 - rd.w0)), dim3(256), 0, 0, (int)rd.w0, (int)rd.w1, pl.d_verts, rd.d_base, pl.d_slot_base);
  if (nt == 0) {
    HIP_TRY(hipMemset(rd.d_trp, 0, sizeof(int) * nv1));
    HIP_TRY(hipDeviceSynchronize());
    return GM_OK;
  }
  HIP_TRY(cnt.alloc(nv1));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(int) * nv1, 0));
  const long long kblocks = std::min<long long>(((long long)nv * 8 + 255) / 256, (long long)g->cu_count * (pl.topo ? 8 : 64));
  const int hub0 = pl.topo ? std::max(0, nv - kHubWin) : nv;
  hipLaunchKernelGGL((cb_task_rows_kernel<false>), dim3((unsigned)kblocks), dim3(256), 0, 0, nv, g->d_rp, g->d_col, g->d_edesc, ntask_of.p, pl.topo ? 1 : 0, hub0,
                     cnt.p, nullptr, nullptr, nullptr);
  HIP_TRY(dev_exclusive_sum(tmp, cnt.p, rd.d_trp, nv1));
  HIP_TRY(dev_malloc(&rd.d_tasks, sizeof(CBuildTask) * (size_t)nt));
  HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(int) * nv1, 0));
  hipLaunchKernelGGL((cb_task_rows_kernel<true>), dim3((unsigned)kblocks), dim3(256), 0, 0, nv, g->d_rp, g->d_col, g->d_edesc, ntask_of.p, pl.topo ? 1 : 0, hub0,
                     cnt.p, rd.d_trp, rd.d_base, rd.d_tasks);
  HIP_TRY(hipGetLastError());
  // host chunks: runs of consecutive vertices whose DAG rows fit the stage (longer rows host nothing), costs from the task lists,
  // heavy chunks cut into parts like gm_tch.hip's (a hub hosts 10^5 in-edges)
  ChunkTable &t = rd.host_tab;
  t.target = std::max(64, std::min(pl.target, pl.stage));
  t.allow_split = true;
  t.bit_words = 0;
  t.part_cap = task_part_cap(g, pl.world);
  t.stage_cap = pl.stage;
  t.rf.tct = 1;
  t.rf.skip_lo = pl.stage;
  t.rf.skip_hi = 0x7fffffff;
  t.bitmap_min_deg = 0x7fffffff;
  double bm_ms = 0;
  int rc = build_table_device(g, t, false, bm_ms, rd.d_trp, &rd.d_tasks[0].len, (int)(sizeof(CBuildTask) / sizeof(int)));
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  return GM_OK;
}

int get_clique_plan(gm_graph *g, int rank, int world, int policy, int target, unsigned long long part_cap, CliquePlan **out) {
  {
    std::lock_guard<std::mutex> lk(g->mu);
    for (auto &pl : g->clique_plans)
      if (pl.rank == rank && pl.world == world && pl.policy == policy && pl.target == target) { *out = &pl; return GM_OK; }
  }
  // (built outside the lock: the solvers have ONE caller per handle; get_table below takes the lock itself)
  HIP_TRY(hipSetDevice(g->device));
  CliquePlan pl;
  pl.rank = rank; pl.world = world; pl.policy = policy; pl.target = target;
  pl.stage = g->max_deg <= kStageCapClique ? kStageCapClique : kCbMaxDeg;
  // narrow chunk table (whole rows, matrices of a chunk <= kBitWords words together; the wide and the huge rows left out)
  {
    RowFilter rf;
    rf.skip_clique_wide = clique_wide_min_words();
    rf.only_hi = kCbMaxDeg;  // rows beyond the stage of the build kernel: mine_kernel's arena path (run_pattern)
    const int rc = get_table(g, target, false, kBitWords, part_cap, kStageCapClique, &pl.tabN, rf, kBitmapMinDeg);
    if (rc) return rc;
  }
  {
    const int rc = ensure_edesc(g);  // (the task records read the targets' {start, length} from the edge descriptors)
    if (rc) return rc;
    setup_trace("clique: narrow table + edge descriptors");
  }
  SetupTimer timer;
  PoolScope pool(g);  // (the per-vertex arrays of the rounds)
  ScanTemp tmp;
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  if (!g->wide_valid) {  // once per graph: the wide vertices, longest rows first (select + stable radix sort by row length)
    const int nv = g->nv;
    DevBuf<int> flag, iota, sel, nsel, degs, keyo;
    HIP_TRY(flag.alloc((size_t)std::max(nv, 1)));
    HIP_TRY(iota.alloc((size_t)std::max(nv, 1)));
    HIP_TRY(sel.alloc((size_t)std::max(nv, 1)));
    HIP_TRY(nsel.alloc(1));
    int m = 0;
    if (nv > 0) {
      hipLaunchKernelGGL(wide_flag_kernel, blocks(nv), dim3(256), 0, 0, nv, g->d_rp, clique_wide_min_words(), flag.p, iota.p);
      size_t bytes = 0;
      HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceSelect::Flagged(tmp.buf.p, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
      HIP_TRY(hipMemcpy(&m, nsel.p, sizeof(int), hipMemcpyDeviceToHost));
    }
    g->n_wide = (size_t)m;
    if (m > 0) {
      HIP_TRY(degs.alloc((size_t)m));
      HIP_TRY(keyo.alloc((size_t)m));
      hipLaunchKernelGGL(gather_deg_kernel, blocks(m), dim3(256), 0, 0, m, sel.p, g->d_rp, degs.p);
      HIP_TRY(dev_malloc(&g->d_wide_sorted, sizeof(int) * (size_t)m));
      size_t bytes = 0;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, degs.p, keyo.p, sel.p, g->d_wide_sorted, m));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp.buf.p, bytes, degs.p, keyo.p, sel.p, g->d_wide_sorted, m));
    }
    g->wide_valid = true;
  }
  {  // topological numbering? (decides whether the in-edge tasks stream only the part of N+(u) beyond v)
    bool topo = false;
    const int rc = graph_is_topological(g, &topo);
    if (rc) return rc;
    pl.topo = topo && !gm_sweep_env("GM_CLIQUE_NO_TOPO");  // (GM_CLIQUE_NO_TOPO: A/B, whole lists streamed)
    if (pl.topo) {  // the dense hub core: rows of the wide vertices' matrices are gathered from it (gm_cgather.hip)
      const int rc2 = ensure_core_bitmap(g);
      if (rc2) return rc2;
      if (g->core_state == 1) pl.core_base = g->core_base;
    }
  }
  {  // this rank's share of the narrow table: every world-th chunk of the cost-ordered dequeue list, a contiguous range, or the
     // chunks of a vertex range
    ChunkTable *tb = pl.tabN;
    const long long n = (long long)tb->n;
    long long first = 0, step = 1, count = 0;
    if (policy == GM_PART_VERTEX) {
      const int rc = table_host_views(g, tb);
      if (rc) return rc;
      const long long vlo = (long long)g->nv * rank / world, vhi = (long long)g->nv * (rank + 1) / world;
      auto first_chunk_at = [&](long long v) {
        long long lo = 0, hi = n;
        while (lo < hi) {
          const long long mid = (lo + hi) / 2;
          if (tb->first_vertex[(size_t)mid] < v) lo = mid + 1; else hi = mid;
        }
        return lo;
      };
      first = first_chunk_at(vlo);
      count = first_chunk_at(vhi) - first;
    } else {
      gm_partition((int64_t)n, rank, world, policy, (int64_t *)&first, (int64_t *)&step, (int64_t *)&count);
    }
    pl.n_first = first; pl.n_step = step; pl.n_count = count;
    pl.d_order = (policy == GM_PART_ROUND_ROBIN) ? tb->d_order[pl.order_which] : nullptr;
  }
  // this rank's share of the wide list
  int64_t wfirst = 0, wstep = 1, wcount = 0;
  gm_partition((int64_t)g->n_wide, rank, world, policy == GM_PART_VERTEX ? GM_PART_RANGE : policy, &wfirst, &wstep, &wcount);
  std::vector<int> hd((size_t)wcount);
  if (wcount > 0) {
    DevBuf<int> degs;
    HIP_TRY(dev_malloc(&pl.d_verts, sizeof(int) * (size_t)wcount));
    HIP_TRY(dev_malloc(&pl.d_slot_base, sizeof(unsigned long long) * (size_t)wcount));
    HIP_TRY(degs.alloc((size_t)wcount));
    hipLaunchKernelGGL(wide_share_kernel, blocks(wcount), dim3(256), 0, 0, (int)wcount, (long long)wfirst, (long long)wstep, g->d_wide_sorted, g->d_rp, pl.d_verts, degs.p);
    pl.verts.resize((size_t)wcount);
    HIP_TRY(copy_to_host(pl.verts.data(), pl.d_verts, sizeof(int) * (size_t)wcount));
    HIP_TRY(copy_to_host(hd.data(), degs.p, sizeof(int) * (size_t)wcount));
  }
  setup_trace("clique: wide share to the host");
  // rounds within the arena budget: the narrow chunks first (in dequeue order), then the wide vertices (longest rows first)
  unsigned long long arena_mb = GM_WIDE_ARENA_MB;
  if (const char *e = gm_opt("GM_WIDE_ARENA_MB")) arena_mb = std::max(1ll, atoll(e));  // (tests: force several rounds)
  const unsigned long long budget_words = (arena_mb << 20) / 4ull;
  // (the per-chunk words come to the host only when the narrow chunks alone overflow the arena: one round needs their sum)
  std::vector<unsigned long long> cw;
  unsigned long long cw_total = 0;
  if (pl.n_count > 0) {
    DevBuf<unsigned long long> dcw, dsum;
    HIP_TRY(dcw.alloc((size_t)pl.n_count));
    HIP_TRY(dsum.alloc(1));
    HIP_TRY(hipMemsetAsync(dsum.p, 0, 8, 0));
    hipLaunchKernelGGL(cb_chunk_words_kernel, blocks(pl.n_count), dim3(256), 0, 0, pl.n_count, pl.n_first, pl.n_step, pl.d_order, pl.tabN->d, g->d_rp, dcw.p);
    hipLaunchKernelGGL(sum_u64_kernel, dim3((unsigned)std::min<long long>((pl.n_count + 255) / 256, 1024)), dim3(256), 0, 0, pl.n_count, dcw.p, dsum.p);
    HIP_TRY(hipMemcpy(&cw_total, dsum.p, 8, hipMemcpyDeviceToHost));
    if (cw_total > budget_words) {
      cw.resize((size_t)pl.n_count);
      HIP_TRY(copy_to_host(cw.data(), dcw.p, sizeof(unsigned long long) * (size_t)pl.n_count));
    }
  }
  setup_trace("clique: words of the narrow chunks to the host");
  std::vector<int> cls_slots, mcls_slots;
  {
    long long c0 = 0;
    size_t s0 = 0;
    while (c0 < pl.n_count || s0 < (size_t)wcount || pl.rounds.empty()) {
      CliqueRound rd;
      unsigned long long words = 0;
      long long c1 = c0;
      if (cw.empty()) {  // every narrow chunk fits one round: the first
        if (c0 < pl.n_count) words = cw_total;
        c1 = pl.n_count;
      } else {
        for (; c1 < pl.n_count; ++c1) {
          if ((c1 > c0) && words + cw[(size_t)c1] > budget_words) break;
          words += cw[(size_t)c1];
        }
      }
      size_t s1 = s0;
      std::vector<int> by_mcls[3];
      if (c1 == pl.n_count) {  // (wide vertices only once the narrow chunks are placed)
        for (; s1 < (size_t)wcount; ++s1) {
          const int d = hd[s1];
          const unsigned long long w = (unsigned long long)d * (unsigned long long)((d + 31) / 32);
          if ((s1 > s0 || c1 > c0) && words + w > budget_words) break;
          words += w;
          pl.wide_edges += (unsigned long long)d;
          by_mcls[clique_mma_class(d)].push_back((int)s1);
        }
      }
      rd.n_pos0 = c0; rd.n_count = c1 - c0;
      rd.w0 = s0; rd.w1 = s1;
      for (int c = 0; c < 3; ++c) {
        rd.mcls_begin[c] = mcls_slots.size();
        mcls_slots.insert(mcls_slots.end(), by_mcls[c].begin(), by_mcls[c].end());
      }
      rd.mcls_begin[3] = mcls_slots.size();
      pl.rounds.push_back(std::move(rd));
      c0 = c1;
      s0 = s1;
      if (c0 >= pl.n_count && s0 >= (size_t)wcount) break;
    }
  }
  setup_trace("clique: rounds (host loop)");
  HIP_TRY(dev_malloc(&pl.d_mcls_slots, sizeof(int) * std::max<size_t>(mcls_slots.size(), 1)));
  if (!mcls_slots.empty()) HIP_TRY(copy_to_device(pl.d_mcls_slots, mcls_slots.data(), sizeof(int) * mcls_slots.size()));
  setup_trace("clique: wide list, classes, rounds (host)");
  unsigned long long need_words = 0;
  for (size_t r = 0; r < pl.rounds.size(); ++r) {
    const int rc = build_clique_round(g, pl, pl.rounds[r], tmp);
    if (rc) { free_clique_plan(pl); return rc; }
    if (pl.core_base >= 0) {
      int rc2 = ensure_core_tri(g);
      if (rc2 == GM_OK) rc2 = build_gather_index(g, pl, pl.rounds[r], tmp);
      if (rc2) { free_clique_plan(pl); return rc2; }
    }
    need_words = std::max(need_words, pl.rounds[r].words);
  }
  if (!mcls_slots.empty()) {  // (the slot offsets are known now: the records the count kernels read behind their dequeue)
    HIP_TRY(dev_malloc(&pl.d_mcls_rec, sizeof(int4) * mcls_slots.size()));
    hipLaunchKernelGGL(mcls_rec_kernel, blocks((long long)mcls_slots.size()), dim3(256), 0, 0, (int)mcls_slots.size(), pl.d_mcls_slots, pl.d_verts, g->d_rp,
                       pl.d_slot_base, pl.d_mcls_rec);
  }
  setup_trace("clique: rounds built");
  const size_t need = (size_t)need_words * 4;
  if (need > g->wide_mat_bytes) {
    if (g->d_wide_mat) dev_free(g->d_wide_mat);
    g->d_wide_mat = nullptr;
    g->wide_mat_bytes = 0;
    HIP_TRY(dev_malloc(&g->d_wide_mat, std::max<size_t>(need, 16)));
    g->wide_mat_bytes = need;
  }
  if (!g->d_wide_queue) HIP_TRY(dev_malloc(&g->d_wide_queue, 65536));
  setup_trace("clique: arena");
  HIP_TRY(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g->mu);
  g->clique_plans.push_back(std::move(pl));
  *out = &g->clique_plans.back();
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// Scheduler policy as index arithmetic on chunk ids (replaces the per-GPU COO copies of
// Scheduler::round_robin, src/common/scheduler.cc:34-85, and EVEN_SPLIT, src/clique/multigpu.cu:42-44).
extern "C" int gm_partition(int64_t n_chunks, int32_t rank, int32_t world, int32_t policy, int64_t *first, int64_t *step,
                            int64_t *count) {
  if (!first || !step || !count || n_chunks < 0) return GM_ERR_INVALID;
  if (world < 1) world = 1;
  if (rank < 0 || rank >= world) return GM_ERR_INVALID;
  if (policy == GM_PART_RANGE || policy == GM_PART_VERTEX) {
    const long long lo = n_chunks * rank / world, hi = n_chunks * (rank + 1) / world;
    *first = lo;
    *step = 1;
    *count = hi - lo;
  } else {
    *first = rank;
    *step = world;
    *count = n_chunks > rank ? (n_chunks - rank + world - 1) / world : 0;
  }
  return GM_OK;
}

// Host-only view of the task-chunk table (no device needed): used by the CPU-side multi-process tests.
extern "C" int gm_chunk_table(int32_t nv, const int64_t *row_ptr, int32_t chunk, int32_t for_clique, int32_t *recs,
                              int64_t cap, int64_t *n_out) {
  if (!row_ptr || !n_out || nv < 0 || (cap > 0 && !recs)) return GM_ERR_INVALID;
  std::vector<int> rp;
  int rc = convert_offsets(row_ptr, nv, row_ptr[nv], rp);
  if (rc) return rc;
  int target = chunk > 0 ? chunk : kDefaultChunk;
  target = std::max(64, std::min(target, kStageCap));
  std::vector<ChunkRec> out;
  unsigned long long mb = 0;
  build_chunks(rp, nv, target, !for_clique, for_clique ? kBitWords : 0, kStageCap, out, mb);  // (the walk the device runs: walk_chunks)
  *n_out = (int64_t)out.size();
  for (int64_t i = 0; i < (int64_t)out.size() && i < cap; ++i) {
    recs[4 * i + 0] = out[i].u_begin;
    recs[4 * i + 1] = out[i].u_end;
    recs[4 * i + 2] = out[i].e_begin;
    recs[4 * i + 3] = out[i].e_end;
  }
  return GM_OK;
}

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_tables_kernel() {}
void gm_touch_tables() { hipLaunchKernelGGL(gm_touch_tables_kernel, dim3(1), dim3(1), 0, 0); }
