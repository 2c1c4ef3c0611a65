// gm_host.h -- host-side internals shared by the translation units behind the C ABI (include/graphminer_amd.h):
//   gm_graph.hip   handle, upload / adopt, orientation, neighbour sort, degree renumbering      (GraphGPU::init, Graph::orientation)
//   gm_tables.hip  task-chunk tables, edge descriptors, task lists, k-clique plans, partitions  (Graph::init_edgelist, Scheduler)
//   gm_launch.hip  the solvers: launch prologue / epilogue, run_pattern, gm_tc / gm_sgl / gm_clique / gm_motif
//   gm_tools.hip   set-op batch, R-MAT generator, PMC / issue-rate calibration kernels, self test
// Host-side counterparts in the reference:
//   GraphGPU::init / init_edgelist          include/graph_gpu.h:69-194
//   launch sizing                            src/triangle/gpu_base.cu:36-45, src/clique/gpu_base.cu:28-50
//   Scheduler::round_robin                   src/common/scheduler.cc:34-85
//   Graph::orientation                       src/common/graph.cc:233-279
// None of that code is reused: tasks are described by a compact chunk table (16 B per ~256 edges)
// instead of per-GPU COO copies, and the multi-GPU split is index arithmetic on chunk ids.
#pragma once
#include "../../include/graphminer_amd.h"
#include "gm_mine.h"
#include "gm_setops.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace gm;  // (internal header: every unit that includes it lives behind the C ABI)

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
extern thread_local std::string g_last_error;
extern thread_local int g_last_hip_error;  // the hipError_t of the most recent hip_fail of this thread (out of memory: optional structures are skipped)
int hip_fail(hipError_t e, const char *what, const char *file, int line);
#define HIP_TRY(call)                                              \
  do {                                                             \
    hipError_t _e = (call);                                        \
    if (_e != hipSuccess) return hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

// The temporaries of the setup paths (a table build has ~20 of them) come out of ONE device allocation per handle: a bump allocator
// that a PoolScope opens at the top of a setup function and rewinds at its end. hipMalloc / hipFree cost 0.1-0.3 ms each and hipFree
// synchronises the device: round 2 measured ~10 ms of allocation overhead per table. What does not fit the pool
// falls through to hipMalloc (and tells the pool how much to ask for next time).
struct TempPool {
  char *base = nullptr;
  size_t cap = 0, off = 0, want = 0;  // want: bytes the open scopes would have liked (pool + fall-through allocations)
  // A handle family (a graph + its cached orientation and renumbered copies) shares the root's pool, and two threads may work on two
  // members at once under their own handle locks.  The pool belongs to ONE thread at a time: the first PoolScope claims it,
  // nested scopes of that thread share it (stack discipline), and a scope opened by another thread meanwhile gets no pool -- its
  // DevBufs fall through to hipMalloc.  No lock is held across a scope, so there is no lock order to get wrong.
  std::mutex claim_mu;
  std::thread::id owner;
  int depth = 0;
};
extern thread_local TempPool *g_temp_pool;  // the pool of the innermost open PoolScope of this thread (gm_graph.hip)

// Temporaries that do not fit the pool (the GB-sized scratch copies of orientation / renumbering / the key stream on a large graph) are
// kept in a small per-device CACHE (a budget of 16 GB and 64 blocks PER DEVICE) instead of going back to the driver: hipFree of a large buffer costs 0.2 ms on a good day and 30 ms
// per GB on a bad one (scripts/alloc_jitter.hip: p90 of hipFree(1 GB) = 30 ms, in phases), and that -- not a kernel -- was the 441 ms
// first call of config 5 the round-4 driver run recorded against the builder's 80 (reproduced in round 5: orientation 88 instead of 7.5
// ms, tables 227 instead of 28, every kernel time unchanged).  A block is handed out again to a request of at least a quarter of its size,
// after a device synchronisation; the cache is emptied when a root handle is freed (gm_graph.hip).
hipError_t big_cache_get(void **p, size_t bytes, size_t *block_bytes);  // a cached block or a fresh hipMalloc
void big_cache_put(void *p, size_t block_bytes);                         // synchronises the device; beyond the cache's budget: hipFree
void big_cache_trim();                                                   // hipFree of every cached block of the current device
constexpr size_t kBigCacheMinBytes = (size_t)4 << 20;
// EVERY device allocation of the library goes through here: hipMalloc, and when the device is out of memory the cached
// temporaries of this device go back to the driver and the request is made once more -- the cache may park up to its budget of memory
// that nothing else can reach, and a setup path must not fail (or quietly drop an optimisation) while it does.
hipError_t dev_malloc_bytes(void **p, size_t bytes);
void dev_free(void *p);           // the counterpart: a block of >= kBigCacheMinBytes goes to the device's cache, anything else to hipFree
void dev_handle_born(int device);  // (gm_graph.hip: the cache lives as long as the device has a handle)
bool dev_handle_died(int device);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_bytes(reinterpret_cast<void **>(p), bytes); }

template <class T>
struct DevBuf {  // RAII device array; pooled when a PoolScope is open and the pool has room
  T *p = nullptr;
  size_t n = 0;
  bool pooled = false;
  size_t block_bytes = 0;  // > 0: a block of the large-temporary cache (returned there by drop())
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { drop(); }
  void drop() {
    if (p && !pooled) {
      if (block_bytes) big_cache_put(p, block_bytes);
      else dev_free(p);
    }
    p = nullptr;
    pooled = false;
    block_bytes = 0;
  }
  // keep = true: the array outlives the scope (release() hands it to a long-lived owner): never from the pool
  hipError_t alloc(size_t count, bool keep = false) {
    drop();
    n = count;
    const size_t bytes = (sizeof(T) * std::max<size_t>(count, 1) + 255) & ~(size_t)255;
    TempPool *pool = keep ? nullptr : g_temp_pool;
    if (pool) {
      pool->want += bytes;
      if (pool->off + bytes <= pool->cap) {
        p = reinterpret_cast<T *>(pool->base + pool->off);
        pool->off += bytes;
        pooled = true;
        return hipSuccess;
      }
    }
    if (!keep && bytes >= kBigCacheMinBytes) return big_cache_get(reinterpret_cast<void **>(&p), bytes, &block_bytes);
    return dev_malloc(&p, bytes);
  }
  // (a pooled array cannot be handed on: alloc(..., true); a cached block can -- it is an allocation of its own, only larger than asked for)
  T *release() { T *q = pooled ? nullptr : p; p = nullptr; n = 0; pooled = false; block_bytes = 0; return q; }
};

struct ScanTemp {  // temp storage of the hipCUB calls, grown on demand
  DevBuf<char> buf;
  hipError_t reserve(size_t bytes) { return bytes <= buf.n ? hipSuccess : buf.alloc(bytes); }
};


// (dev_exclusive_sum, the hipCUB scan over a ScanTemp: gm_scan.h -- only the units that include hipCUB pay for it)

// estimated work (adjacency entries touched) above which a chunk is cut into parts
// Rows longer than kBitmapMinDeg get a dense vertex-id bitmap, longest first, within kBitmapBudget bytes: SPLIT chunks
// probe their own row's, pass Y probes the searched row's (one load instead of ~lg d dependent ones). A probe is a random
// 64 B line, so it only beats a bisection whose upper levels sit in cache when the row is long. Measured (diamond ms on
// R-MAT-20 / -22 / -24, 3-motif on R-MAT-24): min degree 1024: - / 117 / - / 1037;  2048: 25.8 / 94.8 / - / 1068;
// 4096: 30.6 / 94.5 / 1539 / 1057;  8192: 30.8 / 89.6 / 1528 / 1073;  16384: 44.7 / 89.8 / 1528 / 1072;
// 2048 within 256 MB (Infinity-Cache sized): 25.7 / 87.1 / 2275 / 1373;  no pass-Y bitmaps at all: - / 124 / - / 1473.
#ifndef GM_BITMAP_MIN_DEG
#define GM_BITMAP_MIN_DEG 2048
#endif
#ifndef GM_BITMAP_BUDGET_MB
#define GM_BITMAP_BUDGET_MB 8192
#endif
constexpr int kBitmapMinDeg = GM_BITMAP_MIN_DEG;
constexpr unsigned long long kBitmapBudget = (unsigned long long)GM_BITMAP_BUDGET_MB << 20;
constexpr unsigned long long kPartCostCap = 8ull << 20;     // DAG patterns: staged chunks stay whole
#ifndef GM_PART_CAP_SYM
#define GM_PART_CAP_SYM (2ull << 20)
#endif
constexpr unsigned long long kPartCostCapSym = GM_PART_CAP_SYM;  // symmetric-graph patterns (measured on R-MAT-20/22/24: 1 M best for diamond, 2 M for 3-motif)
constexpr int kDefaultChunk = 1024;  // task edges per chunk when the caller does not say

// Which rows a chunk table covers. The k-clique table leaves the wide vertices to the two-phase path; the tables of the
// symmetric-graph patterns come in three: the general one skips the rows of the big-LDS classes, each class table keeps only
// its own window of row lengths.
struct RowFilter {
  int skip_clique_wide = 0;                    // > 0: leave out the vertices with clique_is_wide(d, this many matrix words)
  int skip_lo = 0, skip_hi = 0;                // rows with skip_lo < d <= skip_hi are left out (0, 0 = none)
  int only_lo = -1, only_hi = 0x7fffffff;      // rows with only_lo < d <= only_hi are kept
  int tct = 0;                                 // 1: chunk costs from the task lists of gm_tch.hip (the tasks a chunk hosts, not its own entries)
  __host__ __device__ bool skips(int d) const {
    return (skip_clique_wide > 0 && clique_is_wide(d, skip_clique_wide)) || (d > skip_lo && d <= skip_hi) || !(d > only_lo && d <= only_hi);
  }
  bool operator==(const RowFilter &o) const {
    return skip_clique_wide == o.skip_clique_wide && skip_lo == o.skip_lo && skip_hi == o.skip_hi && only_lo == o.only_lo && only_hi == o.only_hi && tct == o.tct;
  }
};

// A rank's SHARE of a chunk table (world > 1): the records it dequeues, in dequeue order.  All PARTS of a chunk go to the same rank:
// a part is "every nparts-th batch of the chunk's task list", and the order of a host's tasks is the order in which the placement's
// atomics arrived -- different in every process.  Two ranks that split the parts of one chunk would each cut THEIR OWN order of its
// tasks, and the union would not be the chunk (round 5: bench.py --gpus 2 at full size counted 750,563,783 triangles for 750,506,260;
// the in-process rank-share tests of rounds 2 - 4 shared one handle and could not see it).
struct ShareOrder {
  int world = 1, rank = 0, policy = 0, which = -1;  // which: the dequeue order it was cut from (-1: plain record order)
  int *d = nullptr;
  long long n = 0;
  unsigned long long edges = 0;  // task edges of the share
};

struct ChunkTable {
  int target;       // T: CSR entries per chunk
  bool allow_split; // rows longer than the staging capacity may be cut across chunks
  int bit_words;    // clique: LDS bit-matrix budget the chunks were built for (0 = unconstrained)
  unsigned long long part_cap = 0;  // estimated work above which a chunk is cut into parts; 0 = chunks are never cut (nparts == 1 everywhere)
  int stage_cap = 0;                // rows longer than this are SPLIT rows
  RowFilter rf;                     // the rows the table covers
  int bitmap_min_deg = 0;           // rows longer than this got dense bitmaps (allow_split tables)
  std::vector<unsigned long long> cost;  // estimated work per chunk (after cutting)
  ChunkRec *d = nullptr;
  size_t n = 0;
  int *d_slot = nullptr;                 // per chunk: hub bitmap slot or -1
  int *d_row_slot = nullptr;             // per vertex: bitmap slot or -1
  // dequeue orders of the round-robin policy: [0] chunks well above the mean cost first (heaviest first), the rest in vertex
  // order (single rank); [1] all chunks by estimated cost, descending (rank r of n owns every n-th entry)
  int *d_order[2] = {nullptr, nullptr};
  std::vector<int> order[2];
  bool own_bitmaps = true;               // false: d_bitmaps / d_row_slot belong to a BitmapSet of the graph (shared by its tables)
  unsigned *d_bitmaps = nullptr;         // n_bitmaps x bitmap_words
  size_t n_bitmaps = 0;
  unsigned long long bitmap_words = 0;
  unsigned long long max_bit_words = 0;  // largest nel*stride over chunks that exceed bit_words
  // Host views of the per-chunk scalars: what a SHARE of the table needs (a rank of a larger job: its task edges, the vertex-range policy;
  // the k-clique plan; the chunk dumps).  A device-built table keeps them on the device and table_host_views() fetches them on first use --
  // one rank never does: the first pageable device-to-host copy of a process costs ~10 ms in the runtime, which was the largest single
  // item of a first triangle count (GM_SETUP_TRACE).  total_edges / total_cost are fetched with the table (16 bytes).
  std::vector<unsigned long long> edge_prefix;  // edges owned by chunks [0,i)
  std::vector<int> first_vertex;                // u_begin of every chunk (ascending; SPLIT chunks repeat it)
  bool host_ready = false;
  unsigned long long total_edges = 0, total_cost = 0;
  size_t n_with_cost = 0;  // chunks of non-zero cost (device-built tables): they are the first n_with_cost entries of d_order[1]
  int *d_edges = nullptr, *d_firstv = nullptr;  // device twins of the host views (device-built tables)
  unsigned long long *d_cost = nullptr;
  std::list<ShareOrder> shares;  // the shares handed out so far (get_share_order)
};
int table_host_views(gm_graph *g, ChunkTable *t);  // gm_tables.hip
// the share of rank / world under `policy` (round robin: every world-th CHUNK of the order; range: a contiguous run of chunks), cut from
// dequeue order `which` (0 / 1, or -1 = record order); needs the device twins of the per-record scalars (a device-built table)
int get_share_order(gm_graph *g, ChunkTable *t, int world, int rank, int policy, int which, const ShareOrder **out);

// k-clique (k = 4): the plan of one rank's share (gm_mine.h "level 1 re-hosted"; built by get_clique_plan, gm_tables.hip).
// OWNERS are the vertices whose matrices this rank builds and counts: the vertices of its share of the narrow chunk table + its share
// of the wide list. The matrix arena is bounded (GM_WIDE_ARENA_MB): a share whose matrices need more is processed in ROUNDS; a round
// has its own owners, offsets, task lists (every DAG edge of an owner, at the endpoint that hosts it) and host-chunk table.
struct CliqueRound {
  long long n_pos0 = 0, n_count = 0;       // narrow chunks: positions [n_pos0, n_pos0 + n_count) of the rank's share of the narrow table
  size_t w0 = 0, w1 = 0;                    // wide slots [w0, w1) of the plan's vertex list
  size_t mcls_begin[4] = {0, 0, 0, 0};      // ... per class of the matrix-core count kernel (gm_cmma.hip), in d_mcls_slots
  unsigned long long words = 0;             // arena words of the round
  unsigned long long *d_base = nullptr;     // per vertex: word offset of its matrix (nv + 1; non-owners are empty)
  gm::CBuildTask *d_tasks = nullptr;
  size_t n_tasks = 0;
  int *d_trp = nullptr;                     // task-list offsets per host vertex (nv + 1)
  ChunkTable host_tab;                      // chunks of host vertices (rows that fit the stage), costs from the task lists
  // the blocked gather of the wide rows (gm_cgather.hip cgatherb_kernel): the (vertex, block of core rows) units of this round's wide
  // slots sorted by block, cut into work items of about equal probe counts
  uint4 *d_gunits = nullptr;
  unsigned *d_gtab = nullptr;              // the column tables of the round's slots (16-bit positions, two tiles per dword)
  int2 *d_gitems = nullptr;
  size_t n_gunits = 0, n_gitems = 0;
  unsigned long long gather_table_bytes = 0;  // records + row positions + table dwords the units read by construction (tooling: gm_clique4_gather_info)
};
struct CliquePlan {
  int rank = 0, world = 1, policy = 0, target = 0, order_which = 1;
  int stage = 1024;                  // LDS stage of the build kernel: 1024, or kCbMaxDeg when a longer row hosts
  bool topo = false;                 // every DAG edge goes from a smaller to a larger id: matrices strictly upper triangular
  ChunkTable *tabN = nullptr;        // narrow chunk table (a table of the graph)
  long long n_first = 0, n_step = 1, n_count = 0;  // this rank's share of it: positions first + i * step of the dequeue order
  const int *d_order = nullptr;
  std::vector<int> verts;            // this rank's wide vertices; slot = index here (longest rows first)
  int *d_verts = nullptr;
  unsigned long long *d_slot_base = nullptr;  // slot -> word offset of its matrix inside its round's arena
  int *d_mcls_slots = nullptr;
  int4 *d_mcls_rec = nullptr;        // per entry of d_mcls_slots: {d+, word offset of the matrix (low, high), 0} (gm_cmma.hip)
  int core_base = -1;                // >= 0: the rows of the wide vertices' matrices whose first endpoint is >= core_base are gathered from the
                                     // core bitmap of the graph (gm_cgather.hip); the task lists of the streamed build leave them out
  std::vector<CliqueRound> rounds;
  unsigned long long wide_edges = 0;  // task edges of the wide vertices
};

// Dense vertex-id bitmaps of the hub rows: a property of the graph (which rows: longer than min_deg and not left to a
// big-LDS class), shared by every chunk table that needs them -- tables differ per world size / tuning, the bitmaps do not
// (round 1 rebuilt up to 8 GB of them per table: ADVICE r1).
struct BitmapSet {
  int min_deg = 0;
  RowFilter rf;
  unsigned *d_bitmaps = nullptr;
  int *d_row_slot = nullptr;
  size_t n = 0;
  unsigned long long words = 0;
};

struct gm_graph {
  int device = 0;
  int nv = 0;
  long long ne = 0;
  int max_deg = 0;
  int *d_rp = nullptr;   // int32 offsets, owned
  long long *d_rp64 = nullptr;  // a BIG handle (ne >= 2^31): 64-bit offsets instead -- orientation, formula 3-motif and download only
  int *d_col = nullptr;  // col_idx
  bool own_col = true;
  int *d_symdeg = nullptr;  // a DAG made by gm_graph_orient: the degrees of the symmetric graph it came from (its topological numbering sorts by them)
  int2 *d_edesc = nullptr;  // per CSR entry: {rp[col[e]], degree(col[e])}, built on first use (ensure_edesc)
  int *d_trp = nullptr;     // task lists of the shorter-list-streams triangle count (ensure_tasklists): row offsets (nv + 1)
  int2 *d_tdesc = nullptr;  // ... and per task {rp[partner], d(partner)}
  int *d_tedge = nullptr;   // ... and (edge supports only: ensure_tasklists(g, true)) the task's own DAG entry
  // ... and TASK-MAJOR COPIES of the short lists (ensure_tasklists): d_colk = col followed by the keys of every list of <= GM_TC_INLINE_MAX
  // entries in task order, d_tdesck = the descriptors pointing at the copies.  Consecutive tasks of a host then stream consecutive
  // lines instead of one random 64-byte line (or two) each: what bounds the triangle count on LiveJournal-shaped graphs (mean list 9)
  int *d_colk = nullptr;
  int2 *d_tdesck = nullptr;
  unsigned long long n_inline_keys = 0;
  // nv <= 2^24: instead of descriptors that point at the copies, the copies ARE the task list of the short lists -- a key stream in host
  // order, every key tagged with its host's low 8 bits (GraphView::kst), offsets per host vertex, and the longer lists as their own
  // task lists (d_trpl / d_tdescl).  No descriptor, no row search, no flattening for the short lists: one coalesced load per 64 keys.
  unsigned *d_kst = nullptr;  // the stream itself (ensure_keystream, gm_tables.hip)
  int kst_state = 0;          // 2: this handle cannot have one (ids beyond 24 bits, GM_TC_NO_KEY_STREAM)
  int n_long_tasks = 0;
  int *d_kst_rp = nullptr;
  int *d_trpl = nullptr;
  int2 *d_tdescl = nullptr;
  // the same for the edge supports (ensure_keystream(g, true)): per key the DAG entry it was copied from and the entry of its task's own
  // edge, per longer list the entry of its task's own edge.  They belong to d_kst / d_tdescl when the stream was built with them, else to
  // a second set d_kst2 / d_tdescl2 (same offsets, its own order of arrival).
  int2 *d_kst_et = nullptr;  // per key: {entry it was copied from, entry of its task's own edge}
  int *d_tedgel = nullptr;
  unsigned *d_kst2 = nullptr;
  int2 *d_tdescl2 = nullptr;
  int kst_lmax = 0;  // the limit of a "short" list the stream was built with
  // the rows beyond the stage of the task-list kernels (> 2048 entries; ensure_long_rows): ids and the prefix of their lengths
  int *d_long_rows = nullptr;
  long long *d_long_prefix = nullptr;
  int n_long_rows = -1;               // -1: not looked for yet
  long long long_edges = 0;
  unsigned *d_sup = nullptr;  // edge supports: one counter per DAG entry (gm_sup.hip)
  // ... and the MATCH MASKS of the in-edge tasks with long tails (ensure_sup_masks, gm_tables.hip): per DAG entry / per task the offset of
  // the task's mask in the arena (64-bit words; kNoMask: the task keeps its atomics), the arena itself (written and read by every launch)
  unsigned *d_emoff = nullptr, *d_tmoff = nullptr;
  unsigned long long *d_smask = nullptr;
  unsigned long long smask_words = 0;
  int *d_sup_far_rows = nullptr;  // the rows with tails of more than 64 keys (masks of several words), widest ids first
  int n_sup_far_rows = 0;
  int smask_min_tail = 0;  // (0: kSupMaskMinTail)
  int smask_state = 0;  // 0 unknown, 1 built, 2 not applicable (DAG not topological, arena beyond 2^32 words, GM_SUP_NO_MASKS)
  std::vector<int> h_rp;  // host copy of the offsets, fetched on first use (host_rp): download, k-clique tables, SgL renumbering
  std::list<ChunkTable> tables;  // list: handed-out pointers stay valid
  unsigned long long *d_counters = nullptr;  // [4] + queue word, 64 B
  unsigned *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  static constexpr int kEvRing = 64;  // HIP-event pairs of the most recent launches
  hipEvent_t ev[kEvRing][4] = {};      // [0], [1]: around the launch; [2], [3]: around its hub-corner kernel (gm_ctc.hip) when it has one
  bool ev_corner[kEvRing] = {};
  unsigned long long ev_launches = 0;
  int cu_count = 256;
  gm_graph *dag_cache = nullptr;          // oriented copy, built on demand by gm_motif_formula
  gm_graph *relabel_cache[3] = {nullptr, nullptr, nullptr};  // renumbered copies: by degree ascending / descending, topological (get_relabeled)
  int sorted_state = 0;              // 0 unknown, 1 every row strictly ascending, 2 not (graph_rows_sorted): the solvers refuse such a handle
  int topo_state = 0;                // 0 unknown, 1 every edge goes to a larger id, 2 not (graph_is_topological)
  unsigned long long giant_edges = ~0ull;  // sum of the rows beyond kStageCapBig entries (~0: not computed yet)
  double tri_per_edge = -1.0;        // triangles per entry of this DAG, estimated from a sample (-1: not yet; the edge supports' stream switch)
  double mean_sq_deg = -1.0;         // sum_v d(v)^2 / ne: the mean length of the row an entry sits in (-1: not computed yet; topo_view)
  bool topo_relabel_failed = false;  // the (degree, id) numbering is not topological for this DAG: it runs as given
  const gm_graph *ring_alias = nullptr;
  const gm_graph *ring_extra[2] = {nullptr, nullptr};  // 4-motif: the handles that ran the other sub-launches of the most recent calls (gm_kernel_times adds them)
  int *d_idx0 = nullptr;                  // rectangle: #neighbours below v, and the wedge-block prefix
  unsigned long long *d_wblock_prefix = nullptr;
  unsigned long long n_wblocks = 0;
  int4 *d_rect_tasks = nullptr;          // rectangle by wedge accumulation: task list, counter maps
  unsigned long long n_rect_tasks = 0;
  // rectangle with the heavy centres' counter maps in LDS (round 6, gm_mine.hip rect_lds_kernel): row bounds per range boundary, the
  // (centre, range) tasks, and rect_acc_kernel's list with the centres that keep only their ends below `rect_cut` first
  int *d_rect_bnd = nullptr;
  int2 *d_rect_lds_tasks = nullptr;
  int4 *d_rect_cut_tasks = nullptr;
  unsigned long long n_rect_lds_tasks = 0, n_rect_cut_tasks = 0, n_rect_cut = 0;
  int rect_cut = 0;
  gm::RectLdsRanges rect_ranges;
  bool rect_lds_ready = false;
  unsigned *d_rect_acc = nullptr;
  size_t rect_acc_bytes = 0;
  unsigned *d_house_t = nullptr;         // house by wedge accumulation: per-entry tables, task list, 64-bit maps
  unsigned *d_house_tlt = nullptr;
  int4 *d_house_tasks = nullptr;
  unsigned long long n_house_tasks = 0;
  unsigned long long *d_house_acc = nullptr;
  size_t house_acc_bytes = 0;
  int *d_house_touched = nullptr;
  // house with the heavy centres' maps in LDS (round 6, gm_mine.hip house_lds_kernel): as for the rectangle
  int *d_house_bnd = nullptr;
  int2 *d_house_lds_tasks = nullptr;
  int4 *d_house_cut_tasks = nullptr;
  unsigned long long n_house_lds_tasks = 0, n_house_cut_tasks = 0, n_house_cut = 0;
  gm::HouseLdsRanges house_ranges;
  bool house_lds_ready = false;
  int *d_pent_touched = nullptr;         // pentagon by wedge accumulation: touched-vertex lists (same shape as d_rect_acc)
  size_t pent_touched_bytes = 0;
  unsigned long long *d_house_prefix = nullptr;  // house: per-entry task-block prefix
  unsigned long long n_house_blocks = 0;   // handle whose event ring holds this handle's most recent launch
  unsigned long long sum_c2 = 0;          // sum_v C(d(v),2)
  bool sum_c2_valid = false;
  // k-clique: the wide DAG vertices (clique_is_wide), longest rows first, and the per-(rank, world, policy) plans of their two
  // phases (row-group chunks of phase 1, count classes of phase 2, matrix offsets, arena rounds)
  std::list<BitmapSet> bitmap_sets;
  int *d_wide_sorted = nullptr;
  size_t n_wide = 0;
  bool wide_valid = false;
  std::list<CliquePlan> clique_plans;
  unsigned *d_wide_mat = nullptr;      // matrix arena (largest round so far)
  size_t wide_mat_bytes = 0;
  unsigned *d_wide_queue = nullptr;    // dequeue words of the wide launches of one call (zeroed per call)
  // k-clique on a topologically numbered DAG: dense adjacency bitmap of its LAST core_h vertices (the hubs, when the numbering is by
  // degree) -- row v - core_base holds bit w - core_base for every edge v -> w; the rows of the wide vertices' matrices whose first
  // endpoint lies in this core are GATHERED from it (gm_cgather.hip) instead of being built by streamed intersections
  unsigned *d_core = nullptr;
  int core_h = 0, core_base = 0;       // core_h = 0: not built / not applicable (ensure_core_bitmap)
  int core_state = 0;                  // 0 unknown, 1 built, 2 not applicable
  // ... and the same bits cut into BLOCKS of consecutive rows for the blocked gather (gm_cgather.hip cgatherb_kernel; ensure_core_tri)
  unsigned *d_cg_tri = nullptr;        // block images: of every core row the words from its diagonal word on
  int *d_cg_rowbase = nullptr;         // per core row: offset inside its block's image - first stored word
  int *d_cg_bid = nullptr;             // per core row: its block
  int4 *d_cg_blk = nullptr;            // per block: {first row, rows, image offset, image words}
  int n_cg_blocks = 0;
  int cg_tri_state = 0;                // 0 unknown, 1 built, 2 not applicable (no core bitmap, no room)
  // triangle count: the out-edges of the last tc_core_h vertices are counted on the matrix cores from the corner of d_core (gm_ctc.hip) and
  // the key stream / the longer lists of this handle hold no task of a row >= kst_skip_from (ensure_keystream); 0 / nv: no such corner
  int tc_core_h = 0;
  int kst_skip_from = 0x7fffffff;
  long long tc_core_edges = 0;         // entries of the rows >= kst_skip_from
  // the same for the TASK LISTS (ensure_tasklists: the edge supports, the triangle count without a key stream): no task of a row >= tl_skip_from;
  // the supports of the corner's edges come from the symmetric corner d_csym (tl_core_h / 32 words per row) with d_cfirst[row][word] = position
  // of the word's first entry inside its row (ensure_sup_corner, gm_tables.hip; core_tc_block_kernel<true>, gm_ctc.hip)
  int tl_core_h = 0;
  int tl_skip_from = 0x7fffffff;
  unsigned *d_csym = nullptr;
  unsigned short *d_cfirst = nullptr;
  hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};  // class kernels that cannot fill the chip run beside the others (run_pattern)
  hipEvent_t aux_done[3] = {nullptr, nullptr, nullptr};
  TempPool pool;  // temporaries of the setup paths (PoolScope)
  bool counted = false;            // finish_handle has registered the handle with its device (dev_handle_born)
  gm_graph *pool_owner = nullptr;  // a derived handle (renumbered copy, cached orientation) borrows its owner's pool: the owner outlives it,
                                   // and one caller works on a handle family at a time -- a pool of its own is ~0.8 ms of hipMalloc
  gm_setup_times setup = {0, 0, 0, 0, 0};  // accumulated pre-processing time of this handle (gm_graph_setup_times)
  std::mutex mu;
  std::mutex dag_mu;  // guards the lazy creation of dag_cache (ensure_dag_cache, gm_launch.hip)
};

// Opens the handle's temp pool for the DevBufs of the enclosing setup function (nested scopes share it, stack discipline). The pool is
// ONE allocation per handle, made on first use and kept: 24 bytes per vertex + 16 MB, at most 256 MB -- room for the O(nv) and
// O(chunks) arrays of a table build; the sort buffers of a renumbering or a task-list build (O(ne), once per graph) fall through to
// hipMalloc. (A pool sized for everything and regrown per scope was measured first: allocating hundreds of MB per table cost more
// than the twenty small hipMallocs it replaced -- diamond R-MAT-22 table_ms 11.6 -> 40.)
struct PoolScope {
  gm_graph *g;
  TempPool *prev;
  size_t mark = 0;
  bool mine = false;  // this scope holds (a level of) the claim on the root's pool
  explicit PoolScope(gm_graph *g_) : g(g_), prev(g_temp_pool) {
    while (g->pool_owner) g = g->pool_owner;
    TempPool &pl = g->pool;
    {
      std::lock_guard<std::mutex> lk(pl.claim_mu);
      if (pl.depth == 0 || pl.owner == std::this_thread::get_id()) {
        pl.owner = std::this_thread::get_id();
        ++pl.depth;
        mine = true;
      }
    }
    if (!mine) {  // another thread is inside a scope of this handle family: no pool for this one
      g_temp_pool = nullptr;
      return;
    }
    if (!pl.base && !gm_opt("GM_NO_TEMP_POOL")) {
      const size_t need = std::min<size_t>((size_t)256 << 20, (size_t)24 * ((size_t)g->nv + 1) + ((size_t)16 << 20));
      if (dev_malloc(&pl.base, need) == hipSuccess) pl.cap = need;
      else (void)hipGetLastError();  // (no pool: everything falls through to hipMalloc)
    }
    mark = pl.off;
    g_temp_pool = &pl;
  }
  ~PoolScope() {
    (void)hipDeviceSynchronize();  // nothing may still be reading the temporaries when the next scope reuses them
    if (mine) {
      TempPool &pl = g->pool;
      pl.off = mark;
      std::lock_guard<std::mutex> lk(pl.claim_mu);
      --pl.depth;
    }
    g_temp_pool = prev;
  }
};

// wall-clock stopwatch for the setup accounting (host clock: the steps mix host work, copies and synchronised kernels)
struct SetupTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// Device-to-host copies of the O(chunks) / O(hub rows) arrays of the setup paths.  The FIRST device-to-host copy of 64 KB or more in a
// process costs 8 - 10 ms of one-time initialisation inside the HIP runtime (pageable or pinned destination alike; copies of 16 KB
// do not take that path: 15 us each -- measured with a bare HIP program, profiles/r04/ab_setup_first_call.txt).  Up to 1 MB goes in
// 16 KB pieces, so a first call on a fresh process does not pay for it; larger copies are worth the real path.
inline hipError_t copy_to_host(void *dst, const void *src, size_t bytes) {
  constexpr size_t kPiece = (size_t)16 << 10;
  if (bytes > ((size_t)1 << 20)) return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  for (size_t o = 0; o < bytes; o += kPiece) {
    const hipError_t e = hipMemcpy((char *)dst + o, (const char *)src + o, std::min(kPiece, bytes - o), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

inline hipError_t copy_to_device(void *dst, const void *src, size_t bytes) {  // (the same one-time cost the other way round)
  constexpr size_t kPiece = (size_t)16 << 10;
  if (bytes > ((size_t)1 << 20)) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  for (size_t o = 0; o < bytes; o += kPiece) {
    const hipError_t e = hipMemcpy((char *)dst + o, (const char *)src + o, std::min(kPiece, bytes - o), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// GM_SETUP_TRACE: wall-clock marks of the setup steps on stderr, each after a device synchronisation (where do the milliseconds of a
// first call go: kernels, allocations, copies back?).  Off: one predictable branch.
inline void setup_trace(const char *what) {
  static const bool on = getenv("GM_SETUP_TRACE") != nullptr;
  if (!on) return;
  static std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  const auto t0 = std::chrono::steady_clock::now();
  (void)hipDeviceSynchronize();
  const auto t1 = std::chrono::steady_clock::now();
  fprintf(stderr, "[setup] %-34s %8.3f ms (+ %.3f waiting for the device)\n", what, std::chrono::duration<double, std::milli>(t0 - last).count(),
          std::chrono::duration<double, std::milli>(t1 - t0).count());
  last = std::chrono::steady_clock::now();
}

// scope guard: the enclosed once-per-graph pattern setup (device work included) is charged to setup.other_ms
struct OtherSetupScope {
  gm_graph *g;
  SetupTimer t;
  explicit OtherSetupScope(gm_graph *g_) : g(g_) {}
  ~OtherSetupScope() {
    (void)hipDeviceSynchronize();
    g->setup.other_ms += t.ms();
  }
};

// module warm-up: one no-op kernel per translation unit (each is its own code object, loaded on first launch -- 20 ms for the unit
// with the hipCUB instantiations: without this the first table build / first mining launch of a process pays for it)
void gm_touch_mine();
void gm_touch_mine_wide();
void gm_touch_hrow();
void gm_touch_tch();
void gm_touch_sup();
void gm_touch_cbuild();
void gm_touch_cmma();
void gm_touch_cgather();
void gm_touch_ctc();
void gm_touch_sgl();
void gm_touch_tables();
void gm_touch_launch();
void gm_touch_tools();

// ---- across translation units ---------------------------------------------------------------------------------------------
int finish_handle(gm_graph *g);                                 // gm_graph.hip: counters, events, CU count of a new handle
int host_rp(gm_graph *g, const std::vector<int> **out);          // gm_graph.hip: host copy of the offsets, fetched on first use
int convert_offsets(const int64_t *rp64, int nv, long long ne, std::vector<int> &out);  // gm_graph.hip: host-side narrowing / validation
int get_relabeled(gm_graph *g, int mode, gm_graph **out);  // gm_graph.hip: cached renumbered copy (0 / 1 by degree, 2 topological)
int graph_is_topological(gm_graph *g, bool *out);
int graph_rows_sorted(gm_graph *g, bool *out);
int ensure_long_rows(gm_graph *g);  // (gm_tables.hip) d_long_rows / d_long_prefix: the rows of more than kTctStageMax entries
int ensure_core_tri(gm_graph *g);     // (gm_tables.hip) d_cg_* of a handle with a core bitmap; GM_OK also when not applicable
int ensure_core_bitmap(gm_graph *g);  // (gm_tables.hip) d_core / core_h / core_base of a topologically numbered DAG; GM_OK also when not applicable
void free_tables(gm_graph *g);                                   // gm_tables.hip
int get_table(gm_graph *g, int target, bool allow_split, int bit_words, unsigned long long part_cap, int stage_cap, ChunkTable **out,
              const RowFilter &rf = RowFilter(), int bitmap_min_deg = kBitmapMinDeg);
int ensure_edesc(gm_graph *g);
int ensure_tasklists(gm_graph *g, bool with_edges = false);
int ensure_keystream(gm_graph *g, bool edges, bool *built, bool allow_core = false);  // allow_core: the rows of the hub core may stay out (gm_ctc.hip)
int sup_mask_min_tail(const gm_graph *g);  // (gm_tables.hip) kSupMaskMinTail or the option GM_SUP_MASK_MIN when the handle's masks were laid out
int ensure_sup_corner(gm_graph *g);  // (gm_tables.hip) d_csym / d_cfirst of a handle whose task lists leave a hub corner out
int ensure_sup_masks(gm_graph *g);  // (gm_tables.hip) d_emoff / d_tmoff / d_smask of a topologically numbered DAG with task lists; GM_OK also when not applicable
int ensure_mean_sq_deg(gm_graph *g);
int ensure_tri_per_edge(gm_graph *g);  // (gm_launch.hip) |N+(u) ^ N+(v)| averaged over a sample of the DAG's entries
unsigned long long task_part_cap(gm_graph *g, int world);
int clique_wide_min_words();
int get_clique_plan(gm_graph *g, int rank, int world, int policy, int target, unsigned long long part_cap, CliquePlan **out);
void free_clique_plans(gm_graph *g);
int run_pattern(gm::Pattern pat, const gm_graph *cg, const gm_launch *la, int k, uint64_t *h_out, int nout, gm_stats *st, int fin_mode = -1,
                unsigned long long fin_base = 0, unsigned *sup_out = nullptr);  // gm_launch.hip (sup_out: PAT_SUPPORT_PART's buffer)
