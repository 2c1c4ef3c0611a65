// gm_mine.h -- shared declarations between the host API (gm_api.hip) and the mining kernels
// (gm_mine.hip, gm_sgl.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// DEVELOPER OPTIONS.  The shipped library reads NO algorithm switch from the environment (a drop-in behind someone
// else's main must not change what it runs on an ambient variable).  What tests need to reach a path on a small graph, and the documented
// fallbacks, are named options set through the C ABI -- gm_dev_option(name, value), include/graphminer_amd.h -- and read with gm_opt():
//   GM_DIAMOND_PER_EDGE, GM_SUP_STREAM / GM_SUP_NO_MASKS / GM_SUP_MASK_MIN (edge supports), GM_TC_CORE_H / GM_SUP_CORE_H (hub corner),
//   GM_BIG_NE, GM_KST_MAX_KEYS, GM_TOPO_MIN_ROW, GM_WIDE_ARENA_MB, GM_TCT_SPLIT_ALWAYS, GM_RECT_LDS_MIN / GM_RECT_LDS_RANGES (limits lowered for tests), GM_ORIENT_TWO_GATHERS,
//   GM_RELABEL_GLOBAL_SORT (the previous setup paths, compared in tests), GM_NO_TEMP_POOL, and the n-GPU runner's GM_FORCE_RCCL_PATH /
//   GM_DIAMOND_SUPPORTS_MAX_WORLD (host/multi.cc).  Only GM_SETUP_TRACE (setup steps on stderr: diagnostics, no algorithm) is an
//   environment variable.  In -DGM_DEVEL builds (make DEVEL=1) an option that is not set falls back to the environment, and the
// SWEEP switches (tile counts, workgroups per CU, thresholds, rejected variants: the A/B runs recorded under profiles/) exist only there:
// gm_sweep_env() is a constant nullptr otherwise and the branches behind it fold away.
const char *gm_opt(const char *name);  // (gm_graph.hip) the value of a developer option or nullptr
inline const char *gm_sweep_env(const char *name) {
#ifdef GM_DEVEL
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

namespace gm {

// ---- compile-time geometry of one workgroup ---------------------------------------------------
constexpr int kWavesPerBlock = 4;         // 256-thread workgroups: the 4 waves share a chunk's LDS stage, take batches independently
constexpr int kStageCap = 1024;           // adjacency entries a workgroup stages in LDS per chunk (4 KB)
#ifndef GM_STAGE_WIDE
#define GM_STAGE_WIDE 3072
#endif
constexpr int kStageCapWide = GM_STAGE_WIDE;       // ... for the patterns that run on SYMMETRIC graphs (see stage_cap_of)
// workgroup classes of the symmetric-graph patterns: rows of kClassRowMin+1 .. kStageCapMid entries are class 1, rows up to
// kStageCapBig class 2 (hashed sets in LDS, gm_hrow.hip; sorted copies, MineCfg in gm_chunk.h, where the ids are too wide), longer
// rows are cut into pieces of kStageCapBig entries (giant_kernel) -- or stay SPLIT rows with dense bitmaps on the general path.
// (8191, not 8192: the branch-free bisection reads up to 2^bitlen(row) - 2 entries past the row start, which must stay inside
// the workgroup's LDS)
constexpr int kStageCapMid = 8191;
constexpr int kStageCapBig = 24576;
constexpr int kMaxChunkVerts = 256;       // rows per task chunk (local row_ptr slice in LDS)
constexpr int kMarkWindow = 512;          // flattened positions resolved per owner-mark window
#ifndef GM_TILES
#define GM_TILES 2  // 64-wide tiles resolved together per wave in the FILTERED pass; swept on MI355X (2 beats 4 there)
#endif
#define GM_TILES_DEFAULT GM_TILES
#ifndef GM_FILTER_LOG2
#define GM_FILTER_LOG2 15
#endif
constexpr int kFilterLog2 = GM_FILTER_LOG2;           // hashed membership filter: 2^15 bits (4 KB) per workgroup (2^14 and 2^16 measured slower)
constexpr int kFilterBits = 1 << kFilterLog2;
constexpr int kFilterWords = kFilterBits / 32;
constexpr int kQueueCap = 64 * (GM_TILES_DEFAULT + 1);  // candidate queue entries per wave (63 left over + kTiles tiles)
#ifndef GM_BIT_WORDS
#define GM_BIT_WORDS 2048
#endif
constexpr int kBitWords = GM_BIT_WORDS;  // clique: LDS words for the per-chunk adjacency bit-matrix (8 KB)
static_assert(GM_BIT_WORDS >= 2048 && GM_BIT_WORDS % 64 == 0, "rows of up to 2048 columns (stride 64) need 32 of them in one LDS group / tile");

// Task chunk = a contiguous vertex range [u_begin,u_end) and the CSR entries [e_begin,e_end) it owns.
// Normal chunks own whole rows (e_begin == rp[u_begin], e_end == rp[u_end]); a row longer than the
// staging capacity is cut into SPLIT chunks (u_end == u_begin+1, [e_begin,e_end) inside the row).
// A chunk whose estimated work is far above the mean is additionally cut into PARTS: every part stages the whole chunk but
// takes only the 64-edge batches b with b % nparts == part, so that several workgroups share one heavy chunk.
// Batches are 64 task edges, except in SPLIT chunks of the symmetric-graph patterns: there every edge streams a whole list
// against a hub row and 64 hub-hub edges are milliseconds of work for one wave, so the batch is kSplitBatch edges (the
// flattened pass still fills all 64 lanes with keys). SPLIT chunks whose partner lists are short keep 64-edge batches: there the
// per-batch setup (a dozen dependent loads per edge) would dominate -- chunk timings, R-MAT-24 3-motif: rows of ~7 K entries
// spent 4.5 ns per streamed key with 16-edge batches against 1.2 ns for the rows of 17..46 K entries.
constexpr int kSplitBatchMinKeys = 2048;  // mean streamed keys per task edge from which a SPLIT chunk takes the small batches
#ifndef GM_SPLIT_BATCH
#define GM_SPLIT_BATCH 16
#endif
constexpr int kSplitBatch = GM_SPLIT_BATCH;
struct ChunkRec {
  int u_begin, u_end, e_begin, e_end;
  int part, nparts;  // nparts >= 1
  int batch, pad_;   // task edges per batch (64; kSplitBatch in SPLIT chunks whose edges stream long lists); pad_ > 0: k-clique row
                     // group of the wide slot pad_ - 1 (see "k-clique, wide vertices" below)
};

struct GraphView {
  int nv;
  int ne;
  const int *rp;   // int32 row offsets (nv+1), internal copy of row_ptr
  const int *col;  // col_idx
  // Edge descriptors, one per CSR entry e: {rp[col[e]], d(col[e])} -- where the neighbour list of the entry's destination
  // starts and how long it is. A lane reads the descriptor of ITS task edge with a coalesced 8-byte load instead of
  // gathering rp[v], rp[v+1] from a random 64-byte line: in the short-list regime (LiveJournal: mean list 9) that gather
  // was half of the lines a task edge touches. Takes the place of the reference's COO src/dst lists (8 B per task,
  // include/graph_gpu.h:29-30); nullptr = gather from rp.
  const int2 *edesc = nullptr;
  // Task lists of the shorter-list-streams triangle count (gm_tch.hip): per vertex the descriptors {rp[partner], d+(partner)}
  // of the lists it hosts, trp = their row offsets (nv+1); nullptr = not built.
  const int *trp = nullptr;
  const int2 *tdesc = nullptr;
  const int *tedge = nullptr;  // per task: the DAG entry of its own edge (edge supports, gm_sup.hip); nullptr = not built
  // edge supports, MATCH MASKS (gm_sup.hip): per task the offset -- in 64-bit words -- of its match mask in the mask arena
  // (MineParams::smask), kNoMask = the task reports its streamed edges with atomics; nullptr = no masks in this launch
  const unsigned *tmoff = nullptr;
  // KEY STREAM of the short lists (gm_tch.hip; built with the task lists when nv <= 2^24): the keys of every list of <= GM_TC_INLINE_MAX
  // entries in task order -- hosts ascending -- each tagged with the low 8 bits of its host vertex in bits 24..31; kst_rp = offsets per
  // host vertex (nv + 1).  A chunk of hosts streams ONE contiguous range; trp / tdesc then hold only the longer lists.
  const unsigned *kst = nullptr;
  const int *kst_rp = nullptr;
  const int2 *kst_et = nullptr;  // edge supports: per key of kst {the DAG entry it was copied from, the entry of its task's own edge}
};

// lookup lists of at least this many keys are streamed one task at a time with a wave-uniform descriptor, shorter ones flattened 64 to a batch
// (gm_flat.h, gm_hset.h, gm_tch.hip)
#ifndef GM_LONG_LIST
#define GM_LONG_LIST 192
#endif
constexpr int kLongList = GM_LONG_LIST;
constexpr unsigned kNoMask = 0xffffffffu;
// edge supports: an in-edge task whose tail has at least this many keys reports its streamed edges as a match mask (gm_sup.hip)
#ifndef GM_SUP_MASK_MIN_TAIL
#define GM_SUP_MASK_MIN_TAIL 2
#endif
constexpr int kSupMaskMinTail = GM_SUP_MASK_MIN_TAIL;
constexpr int kSupMaskSpare = 3;  // spare 64-bit words behind the mask of a long list (tile groups of at most kSupMaskSpare + 1 tiles)

enum Pattern : int { PAT_TC = 0, PAT_DIAMOND = 1, PAT_MOTIF3 = 2, PAT_CLIQUE4 = 3, PAT_CLIQUEK = 4 /* k = 5..12 */,
                     PAT_MOTIF4E = 5 /* per-edge sums of the 4-motif formula */,
                     PAT_DAGSTATS = 6 /* tooling: sum n, sum n^2, sum_{matches} d+(w) for the 4-clique algorithmic bytes */,
                     PAT_SUPPORT = 7 /* edge supports from the DAG's triangles + sum C(t, 2): the diamond count (gm_sup.hip) */,
                     PAT_SUPPORT_PART = 8 /* ... a rank's share of the supports only, into the caller's buffer (gm_diamond_support_partial) */,
                     PAT_CLIQUEK_DEEP = 9 /* kernel side only: the instance of the mining kernel PAT_CLIQUEK launches for k = 9..12 (gm_chunk.h) */ };

// Symmetric-graph patterns stage up to 3072 entries: on skewed graphs thousands of rows have 1-3 K neighbours; with a
// 1024-entry stage they are SPLIT rows whose keys are bisected in HBM, otherwise ordinary staged chunks behind
// the LDS filter. Measured (diamond R-MAT-22 / 3-motif R-MAT-24, ms) before SPLIT chunks pre-filtered in LDS: 1024: 94.4 / 1056,
// 2048: 82.5 / 1021, 4096: 77.4 / 920, 8192: 104.9 / 1111; after (diamond R-MAT-22 / 3-motif R-MAT-24 / diamond R-MAT-20): 2048: 28.2 / 748 /
// 10.1, 3072 (31.5 KB of LDS, 5 workgroups per CU): 28.2 / 625 / 10.2, 4096 (36 KB, 4 per CU): 27.9 / 706 / 13.8.
// Which endpoint of an undirected edge {u, v} hosts its task in the symmetric-graph patterns (a = d(u), b = d(v)); asked
// from u's side: true = u's row hosts. Any rule that picks exactly one endpoint gives the same counts. Normally the LONGER
// row hosts (it is staged or bitmapped, the shorter list is streamed: min(a, b) keys). kProbeCost > 1 adds an exception for
// hosts too long for the LDS stage, whose keys are verified against a bitmap in HBM: against a partner that fits the stage and
// is less than kProbeCost times shorter, the SHORTER row hosts and streams the long list through its filter. It was needed
// (4) while every streamed key of a SPLIT chunk was an HBM probe (5.2 ns vs 1.25 ns per key and workgroup); since SPLIT chunks
// pre-filter in LDS the two paths cost the same and 1 (no exception) measures best: diamond R-MAT-22 27.9 / 28.6 / 31.8 / 36.4 ms
// for 1 / 2 / 3 / 4.
#ifndef GM_PROBE_COST
#define GM_PROBE_COST 1
#endif
constexpr int kProbeCost = GM_PROBE_COST;
#ifndef GM_MID_PROBE_COST
#define GM_MID_PROBE_COST 1
#endif
constexpr int kMidProbeCost = GM_MID_PROBE_COST;
constexpr int kMidRowMax = 16384;
__host__ __device__ inline bool sym_hosts(int a, int b, int u, int v, int stage_cap) {
  const bool u_longer = (a > b) || (a == b && u > v);
  const int dl = u_longer ? a : b, ds = u_longer ? b : a;
  // kProbeCost applies to every long row; kMidProbeCost only to the rows just above the stage (<= kMidRowMax entries): each of
  // them has its own multi-MB bitmap that only a handful of chunks ever touch, so their probes are cold HBM accesses
  // (chunk timings, R-MAT-24 3-motif: 4.5 ns per key for rows of ~7 K entries, 1.1 ns for rows of 17..46 K, 1.4 ns staged).
  // Measured (diamond R-MAT-22 / 3-motif R-MAT-24 / diamond R-MAT-20, ms): 1: 28.1 / 460 / 10.3, 2: 28.3 / 460 / 10.3, 3: 28.2 / 454 / 11.6,
  // 4: 30.7 / 421 / 12.3 -- no setting wins on all three, the default stays 1 (off).
  const int k = (dl <= kMidRowMax) ? kMidProbeCost : kProbeCost;
  const bool shorter_hosts = dl > stage_cap && ds <= stage_cap && (long long)dl < (long long)k * ds;
  return u_longer != shorter_hosts;
}

// k-clique keeps the 1024-entry stage: rows above it are whole-row chunks searched in HBM, but staging them (1536 entries
// still leave 5 workgroups per CU) measured SLOWER, 257 -> 282 ms on R-MAT-22 ef 28 -- their time is the d x d/32 bit-matrix.
constexpr int kStageCapClique = 1024;
constexpr int stage_cap_of(int pat) {
  return (pat == PAT_DIAMOND || pat == PAT_MOTIF3 || pat == PAT_MOTIF4E) ? kStageCapWide
         : (pat == PAT_CLIQUE4 || pat == PAT_CLIQUEK || pat == PAT_CLIQUEK_DEEP) ? kStageCapClique : kStageCap;
}

struct MineParams {
  GraphView g;
  const ChunkRec *chunks;
  const int *chunk_slot;         // per chunk: bitmap slot of its hub row, or -1 (may be nullptr)
  const int *order;              // dequeue position -> chunk id (nullptr = identity)
  unsigned long long *chunk_ticks;  // diagnostics (-DGM_DEBUG_CHUNKS builds): per dequeue position, wall_clock64 ticks spent
  const unsigned *bitmaps;       // dense vertex-id bitmaps of the longest rows, bitmap_words each
  unsigned long long bitmap_words;
  const int *row_slot;           // per vertex: its bitmap slot or -1 (nullptr when the table has no bitmaps)
  int first, step, count;        // this rank owns chunk ids first + i*step, i in [0,count)
  int grab;                      // chunks taken per dequeue
  unsigned *queue;               // dequeue head (zeroed before launch)
  unsigned long long *counters;  // [4] accumulators (zeroed before launch)
  unsigned *scratch;             // clique: global bit-matrix arena, scratch_words per workgroup
  unsigned long long scratch_words;
  // k >= 5: the slot is k - 2 REGIONS of scratch_region words (the vertex's matrix, then one compacted sub-matrix per deeper level) followed
  // by k - 2 position lists of scratch_plist ints (rows beyond 4096 columns: cliquek_count_sub_any)
  unsigned long long scratch_region;
  int scratch_plist;
  int cost_x_step;  // direction rule: X if b*(xb + xs*lg a) <= a*(yb + ys*lg b)
  int cost_y_step;
  int cost_x_base;
  int cost_y_base;
  int cost_y_bitmap;  // pass Y cost per key when row v has a dense bitmap (one random probe)
  int k;
  unsigned long long *smask;  // edge supports: the match-mask arena (GraphView::tmoff), nullptr = none
  int flags;  // bit 0: never stage adjacency in LDS; bit 3: no hashed filter in front of the LDS bisection; bit 9: ignore hub bitmaps (A/B switches)
};

// nested SgL patterns (gm_sgl.hip)
enum SglPattern : int { SGL_RECTANGLE = 0, SGL_HOUSE = 1, SGL_PENTAGON = 2, SGL_DIAMOND = 3 /* listing (nested) form */ };

struct SglParams {
  GraphView g;
  long long first, step, count;  // this rank owns entry-chunk ids first + i*step, i in [0,count)
  int chunk;                     // CSR entries per task chunk
  unsigned *queue;
  unsigned long long *counters;
  int *scratch;                  // house: max_deg ints per wave
  int max_deg;
};
hipError_t launch_sgl_nested(int pat, const SglParams &p, int grid_blocks, hipStream_t stream);

// flattened rectangle / pentagon (gm_mine.hip): tasks are WEDGES (v1, v0, v2), v2 < v1 < v0, 64 per wave
struct RectParams {
  GraphView g;
  const int *idx0;                         // idx0[v] = number of neighbours of v that are < v
  const unsigned long long *block_prefix;  // prefix sum over v of ceil(C(idx0[v],2) / 64)
  unsigned long long first, step, count;   // this rank owns wedge-block groups first + i*step, i in [0,count)
  unsigned long long nblocks;              // total wedge blocks
  int group;                               // wedge blocks per dequeue
  unsigned long long *queue;               // 64-bit dequeue head
  unsigned long long *counters;
};
hipError_t launch_rect_flat(const RectParams &p, bool pentagon, int grid_blocks, hipStream_t stream);
hipError_t launch_idx0(const GraphView &g, int *idx0, hipStream_t stream);

// rectangle by wedge accumulation (gm_mine.hip): per centre v0 a vertex-indexed counter map (the reference's "c-map"
// idea, src/clique/omp_recursive.cc:59) replaces the per-wedge intersections
struct RectAccParams {
  GraphView g;
  const int *idx0;
  const int4 *tasks;                      // {v0, -2, -2, -2}: heavy centre, all 4 waves; else up to 4 light centres (-1 = none)
  unsigned long long first, step, count;  // this rank owns tasks first + i*step
  unsigned *acc;                          // one zeroed counter map per wave: grid * 4 * acc_stride
  int *touched;                           // one list per wave (same stride): the vertices whose counter is non-zero
  unsigned long long acc_stride;
  unsigned long long *queue;
  unsigned long long *counters;
  // round 6: the first n_cut tasks are centres whose 2-path ends at or above `cut` are counted out of LDS maps (rect_lds_kernel):
  // they walk the ends below cut only (bnd0[x] = where the entries >= cut of row x begin).  n_cut = 0: every end, as before.
  unsigned long long n_cut;
  int cut;
  const int *bnd0;  // bnd0[x * bnd_stride]
  int bnd_stride;
};
hipError_t launch_rect_acc(const RectAccParams &p, int grid_blocks, hipStream_t stream);

// rectangle, the counter maps of the heavy centres in LDS (gm_mine.hip rect_lds_kernel): the ids [cut, nv) of a graph numbered ascending in
// degree are cut into ranges, a task is (centre v0, range k): one workgroup walks the 2-paths v0 - x - w with w in the range and counts them
// in a dense map in LDS.  A counter never exceeds d(w), and the degrees fall quickly below the hubs: a range whose largest degree is below
// 2^8 / 2^16 packs four / two counters into a word (128 K / 64 K ids per range instead of 32 K), so that sixteen ranges reach far down.
#ifndef GM_RECT_LDS_WORDS
#define GM_RECT_LDS_WORDS 32768
#endif
constexpr int kRectLdsWords = GM_RECT_LDS_WORDS;  // 128 KB of counters
constexpr int kRectLdsRanges = 32;    // at most
constexpr int kRectLdsWaves = 16;
struct RectLdsRanges {
  int n;                          // ranges in use
  int rb[kRectLdsRanges + 1];     // range k = ids [rb[k], rb[k + 1]), ascending; rb[0] = cut, rb[n] = nv
  int lb[kRectLdsRanges];         // log2 of the counter width of range k: 5, 4 or 3
};
struct RectLdsParams {
  GraphView g;
  const int *idx0;
  const int2 *tasks;  // {v0, k}: range k of centre v0; {v0, -1}: every range of a centre with at most kRectLdsWaves * 64 neighbours below it
  unsigned long long first, step, count;
  const int *bnd;  // nv x (n + 1): bnd[x * (n + 1) + k] = index into col[] of the first entry >= rb[k] of row x
  RectLdsRanges r;
  unsigned long long *queue;
  unsigned long long *counters;
};
hipError_t launch_rect_lds(const RectLdsParams &p, int grid_blocks, hipStream_t stream);
hipError_t launch_rect_bounds(const GraphView &g, const RectLdsRanges &r, int *bnd, hipStream_t stream);
hipError_t launch_rect_blockmax(const GraphView &g, int *blockmax, hipStream_t stream);  // blockmax[b] (zeroed) = largest degree among the ids [nv - (b + 1) W, nv - b W), W = kRectLdsWords
hipError_t launch_rect_work_cut(const GraphView &g, const int *idx0, const int *bnd0, int bnd_stride, unsigned long long *work, hipStream_t stream);
hipError_t launch_rect_work(const GraphView &g, const int *idx0, unsigned long long *work, hipStream_t stream);

// house by wedge accumulation (gm_mine.hip): per-entry triangle tables + one 2-path walk per centre with a 64-bit
// (weighted sum | count) map; same task list layout as RectAccParams
struct HouseAccParams {
  GraphView g;
  const unsigned *t;    // per CSR entry (v0 -> v1): |N(v0) ^ N(v1)|
  const unsigned *tlt;  // per CSR entry (v0 -> v1): |{x in N(v0) ^ N(v1) : x < v0}|
  const int4 *tasks;
  unsigned long long first, step, count;
  unsigned long long *acc;  // one zeroed 64-bit map per wave: grid * 4 * acc_stride
  int *touched;             // one list per wave (same stride): the vertices whose map entry is non-zero
  unsigned long long acc_stride;
  unsigned long long *queue;
  unsigned long long *counters;
  // round 6: the first n_cut tasks are centres whose 2-path ends at or above `cut` are summed out of LDS maps (house_lds_kernel): phase 1
  // walks their ends below cut only (bnd0[x * bnd_stride] = where the entries >= cut of row x begin).  n_cut = 0: every end, as before.
  unsigned long long n_cut;
  int cut;
  const int *bnd0;
  int bnd_stride;
};
// house, the (count | weighted sum) maps of the heavy centres in LDS (gm_mine.hip house_lds_kernel): ranges of kHouseLdsIds ids with one 64-bit
// word each -- the same packing as the global maps -- from the last id down; tasks as for the rectangle (RectLdsParams)
constexpr int kHouseLdsIds = 16384;  // 128 KB of 64-bit counters
constexpr int kHouseLdsRanges = 256;  // at most: the last 4 M ids (and a row-bound table of at most 4 GB: gm_launch.hip)
struct HouseLdsRanges {
  int n, cut, nv;  // range k = ids [rb(k), rb(k + 1)), ascending; rb(0) = cut, rb(n) = nv
  __host__ __device__ int rb(int k) const {
    const long long b = (long long)cut + (long long)k * kHouseLdsIds;
    return k >= n ? nv : (int)(b < (long long)nv ? b : (long long)nv);
  }
};
struct HouseLdsParams {
  GraphView g;
  const unsigned *t;
  const int2 *tasks;  // {v0, k}: range k of centre v0; {v0, -1}: every range of a centre with at most kRectLdsWaves * 64 neighbours
  unsigned long long first, step, count;
  const int *bnd;  // nv x (n + 1)
  HouseLdsRanges r;
  unsigned long long *queue;
  unsigned long long *counters;
};
hipError_t launch_house_lds(const HouseLdsParams &p, int grid_blocks, hipStream_t stream);
hipError_t launch_house_bounds(const GraphView &g, const HouseLdsRanges &r, int *bnd, hipStream_t stream);
hipError_t launch_house_work_cut(const GraphView &g, const int *bnd0, int bnd_stride, unsigned long long *work, hipStream_t stream);
hipError_t launch_edge_tab(const GraphView &g, unsigned *t, unsigned *tlt, unsigned long long *queue, int grid_blocks, hipStream_t stream);
hipError_t launch_house_acc(const HouseAccParams &p, int grid_blocks, hipStream_t stream);
hipError_t launch_house_work(const GraphView &g, unsigned long long *work, hipStream_t stream);

// pentagon by wedge accumulation (gm_mine.hip): the rectangle counter map + a walk over the touched vertices
struct PentAccParams {
  GraphView g;
  const int *idx0;
  const unsigned *tlt;  // per CSR entry (v0 -> v1): |{x in N(v0) ^ N(v1) : x < v0}|
  const int4 *tasks;
  unsigned long long first, step, count;
  unsigned *acc;   // one zeroed counter map per wave: grid * 4 * acc_stride
  int *touched;    // one list per wave (same stride): the vertices whose counter is non-zero
  unsigned long long acc_stride;
  unsigned long long *queue;
  unsigned long long *counters;
};
hipError_t launch_pent_acc(const PentAccParams &p, int grid_blocks, hipStream_t stream);

// flattened house (gm_mine.hip): tasks are (v0, v1, v3) with v1 < v0 in N(v0), v3 in N(v1) \ {v0}, 64 v3 per wave
struct HouseParams {
  GraphView g;
  const unsigned long long *entry_prefix;  // prefix over CSR entries e of ceil(d(col[e]) / 64) if col[e] < row(e), else 0
  unsigned long long first, step, count;   // this rank owns block groups first + i*step, i in [0,count)
  unsigned long long nblocks;
  int group;
  int no_bits;  // test hook: always take the long-row path (S membership by bisection in N(v1))
  unsigned long long *queue;
  unsigned long long *counters;
};
hipError_t launch_house_flat(const HouseParams &p, int grid_blocks, hipStream_t stream);
hipError_t launch_house_blocks(const GraphView &g, unsigned *nblk, hipStream_t stream);

// ---- k-clique, wide vertices -----------------------------------------------------------------------------------------------------
// A DAG vertex u whose d x d adjacency bit-matrix over N+(u) exceeds the 8 KB LDS budget of a narrow chunk (d+ > 256) is WIDE: its
// matrix is counted by ONE big-LDS workgroup (gm_cmma.hip, clique_mma_kernel) that copies it from the matrix arena into LDS (up to
// 144 KB) and counts sum_{i,j} M_ij (M M^T)_ij there on the matrix cores (rounds 2 - 3: popcounts on the vector ALU, deleted in round 5). (Round 1 kept one arena slot per workgroup, so a wide vertex was built
// AND counted by a single workgroup: 88 % of the kernel time, 382 GB of arena re-reads per launch; round 2 built the rows with a lean
// kernel that streamed N+(v) of every edge; since round 3 every row is built where the LONGER list is staged: gm_cbuild.hip, below.)
constexpr int kWideMaxDeg = 2048;   // wider DAG rows stay on the mining kernel's per-workgroup arena path
// min_words: the matrix size (words) from which a vertex is counted by the big-LDS classes; kBitWords = what no longer fits a narrow chunk
__host__ __device__ inline bool clique_is_wide(int d, int min_words = kBitWords) {
  return (long long)d * ((d + 31) / 32) > min_words && d <= kWideMaxDeg;
}
struct CliqueCountParams {
  const int *rp;
  const int *verts;                 // slot -> vertex (this rank's wide vertices of the round)
  const unsigned long long *base;   // slot -> word offset of the vertex's matrix in `mat`
  const unsigned *mat;              // matrix arena
  const int *slots;                 // the slots of this launch (one count class), heaviest first
  const int4 *qrec;                 // ... and, in the same order, {d+, matrix offset low, high, -}: ONE load behind the dequeue instead of slot -> vertex -> row bounds + offset
  int count;
  unsigned *queue;                  // dequeue head (zeroed before launch; its own word)
  unsigned long long *counters;     // [0] += 4-cliques
  int topo;                         // the DAG is numbered topologically: the matrices are strictly upper triangular
};

// ---- level 1 of the wide vertices GATHERED from the dense bitmap of the hub core (gm_cgather.hip) ------------------------------------
struct CGatherParams {
  const int *rp, *col;
  const int *verts;                 // slot -> vertex (this rank's wide vertices, longest rows first)
  const unsigned long long *base;   // slot -> word offset of the vertex's matrix in `mat`
  unsigned *mat;                    // matrix arena
  const unsigned *core;             // core bitmap: row v - core_base, bit w - core_base, core_words words per row
  int core_base, core_words;
  unsigned long long core_bytes;    // size of the bitmap (< 2^32: one buffer resource addresses it)
  int first_slot, count;            // the slots of this launch: first_slot + q
  unsigned *queue;                  // dequeue head (zeroed before launch)
};
hipError_t launch_cgather(const CGatherParams &p, int grid_blocks, hipStream_t stream);
int cgather_per_cu();

// ---- k-clique (k = 4), first level of the wide vertices, BLOCKED gather (gm_cgather.hip, round 6) -------------------------------------
// The same rows as launch_cgather, the loop nest turned round: a BLOCK of consecutive core rows -- as many as fit kCgbWords words of LDS,
// a row keeping only its words right of the diagonal, so a block is 16 rows at the bottom of the core and ~1000 at its top -- is resident in
// LDS, and the (vertex, rows of the vertex inside the block) UNITS listed for it once per plan are streamed past it: a unit reads its
// vertex's column table (2 bytes per column, coalesced, once for all its rows in the block) instead of 128-byte lines of 4 KB core rows for a
// few probes each.  R-MAT-22 ef 28 (scripts/exp/cgather_blocks.py): 51.7 M rows = 16.5 G probes pull 46 GB of lines; 17.2 M units read ~12 GB.
constexpr int kCgbWords = 16384;   // LDS words of a block image (64 KB: two workgroups per CU)
constexpr int kCgbMaxRows = 1024;  // rows of a block (their LDS offsets: 4 KB)
struct CGatherBParams {
  unsigned *mat;          // matrix arena of the round
  const unsigned *tri;    // the block images one after the other: of every core row the words from its diagonal word on (gm_host.h d_cg_tri)
  const int *rowbase;     // per core row: (word offset of the row inside its block's image) - (index of its first stored word)
  const int4 *blk;        // per block: {first core row, rows, image offset in tri (words, a multiple of 4), image words}
  const unsigned *tab;    // column tables: per vertex, dword 64 T + l = q(128 T + l) | q(128 T + 64 + l) << 16, q = id - (core_base & ~31)
  const uint4 *units;     // {start of the vertex's column table, word offset of its matrix in mat, d | first row index << 16, rows}; by block
  const int2 *items;      // work items: {first unit, block}; items[count].x = number of units
  int delta, count;       // delta = core_base & 31
  unsigned *queue;        // dequeue head (zeroed before launch)
};
hipError_t launch_cgatherb(const CGatherBParams &p, int grid_blocks, hipStream_t stream);
int cgatherb_per_cu();
// (host side of the block geometry, shared by the setup and its tests)
// column bits are counted from the core's base rounded down to a multiple of 32 (delta = core_base & 31 phantom columns in front): the
// word of a column is then (vertex id >> 5) - (base >> 5) on every graph
__host__ __device__ inline int cgb_first_word(int p, int delta) { return (p + delta + 1) >> 5; }  // first stored word of core row p: the one that holds column p + 1
__host__ __device__ inline int cgb_words(int h, int delta) { return (h + delta + 31) >> 5; }       // words of a whole row

// ---- triangle count: the triangles of the hub core on the matrix cores (gm_ctc.hip) ---------------------------------------------------
// The corner of the core bitmap that holds the out-edges of the LAST h vertices, sum_{i,j} M_ij (M M^T)_ij over 64 x 64 blocks of (i, j):
// one wave per block, blocks dealt by a dequeue word, t = first + q * step for a rank's share.
constexpr int kCtcWaves = 4;
constexpr int kCtcMaxH = 32768;        // f32 accumulators stay exact; the core bitmap (kCoreHDefault) is no larger
constexpr int kTcCoreHDefault = 32768; // the largest corner considered (tc_core_size, gm_tables.hip: by density); GM_TC_CORE_H overrides
struct CoreTcParams {
  const unsigned *core;      // the core bitmap (gm_host.h d_core): row_words words per row
  int row_words;
  int row0, word0;           // the corner: first row of the bitmap, first word of its rows
  int h;                     // vertices of the corner
  int ntasks;                // blocks (x pieces of their column range): filled in by launch_core_tc
  int first, step;           // this launch takes the tasks first + q * step (rank, world)
  unsigned *queue;           // dequeue word (zeroed before launch)
  unsigned long long *counters;
  // the edge supports' corner (launch_core_sup): DAG offsets, first vertex of the corner, per (row, word) the position of the word's first
  // entry inside its row, the support array
  const int *rp;
  int base;
  const unsigned short *first_pos;
  unsigned *sup;
};
hipError_t launch_core_tc(CoreTcParams p, int cu_count, hipStream_t stream);
hipError_t launch_core_sup(CoreTcParams p, int cu_count, hipStream_t stream);
hipError_t launch_core_sym_fill(int nv, int base, int words, long long e0, long long e1, const int *rp, const int *col, unsigned *bits,
                                unsigned short *first_pos, int cu_count, hipStream_t stream);
bool core_tc_fast_path(const CoreTcParams &p);

// ---- the same counts on the matrix cores (gm_cmma.hip; the default since round 4, tune[6] & 0x20000: the vector-ALU classes above) ----
// sum_{i,j} M_ij (M M^T)_ij as FP4 MFMA over 64 x 64 blocks of (i, j).  LDS copy: rows padded to a multiple of 64, row stride = the
// (even) block width rounded up to 2 mod 4 words, so that the 64 lanes of an operand fragment (32 rows x 2 adjacent words) read 64
// different banks.  Classes: 0 = whole matrix in 36 KB (d+ <= 512; 4 waves, four workgroups per CU), 1 = whole matrix in 144 KB
// (d+ <= 1024; 16 waves, one per CU), 2 = column blocks of an even number of words (<= 8 per vertex, a queue entry each).
constexpr int kMmaWavesS = 4, kMmaWordsS = 9216;
constexpr int kMmaWavesL = 16, kMmaWordsL = 36864;
__host__ __device__ inline int clique_mma_stride(int cw_even) { return (cw_even & 3) == 2 ? cw_even : cw_even + 2; }
__host__ __device__ inline bool clique_mma_fits(int d, int cw_even, int budget_words) {
  return (long long)((d + 63) & ~63) * clique_mma_stride(cw_even) <= budget_words;
}
// words per column block: the fewest equal blocks (>= 2) whose copy fits
__host__ __device__ inline int clique_mma_block_words(int d, int budget_words) {
  const int stride = (d + 31) / 32;
  for (int nb = 2; nb < 8; ++nb) {
    const int cw = (((stride + nb - 1) / nb) + 1) & ~1;
    if (clique_mma_fits(d, cw, budget_words)) return cw;
  }
  return (((stride + 7) / 8) + 1) & ~1;
}
__host__ __device__ inline int clique_mma_class(int d) {
  const int cw = (((d + 31) / 32) + 1) & ~1;
  if (clique_mma_fits(d, cw, kMmaWordsS)) return 0;
  return clique_mma_fits(d, cw, kMmaWordsL) ? 1 : 2;
}
hipError_t launch_clique_mma(int cls, const CliqueCountParams &p, int grid_blocks, hipStream_t stream);
size_t clique_mma_lds_bytes(int cls);
int clique_mma_threads(int cls);

// ---- k-clique (k = 4), level 1 RE-HOSTED (gm_cbuild.hip) -------------------------------------------------------------------------
// Row i of u's adjacency bit-matrix over N+(u) is N+(u) ^ N+(v), v = N+(u)[i] -- the triangle list of the DAG edge u -> v with
// positions. Like the triangle count of gm_tch.hip it is symmetric in which list is staged: the edge is a TASK of the endpoint with
// the LONGER out-list, whose row sits in LDS, and the other list is streamed (sum min(d+(u), d+(v)) keys instead of sum d+(v)):
//   type A  host = u: N+(v) streamed, a match at position p of the staged N+(u) is bit p of the row;
//   type B  host = v: N+(u) streamed, a match at stream index k is bit bit_off + k of the row -- under a topological numbering of
//           the DAG only the keys beyond v can be in N+(v): the stream starts at position i + 1 (= bit_off).
// Either way the row belongs to u's matrix: every task carries the word offset of ITS row in the MATRIX ARENA (all matrices of the
// launch, d x ceil(d/32) words per vertex with 3 <= d+ <= kCbMaxDeg, in vertex order) and the wave that built it stores it there.
// Counting is a second set of launches over finished matrices: clique_small_kernel (matrices of a narrow chunk copied to LDS) and the
// big-LDS classes S / L / X of gm_wide.hip.
constexpr int kCbMaxDeg = kWideMaxDeg;  // 2048: the longest row that owns a matrix here / hosts tasks (longer rows: mine_kernel's arena path)
constexpr int kCbMinDeg = 3;            // a vertex with fewer out-neighbours is in no 4-clique as its smallest member
#ifndef GM_CB_ROWBUF
#define GM_CB_ROWBUF 256
#endif
constexpr int kCbRowBuf = GM_CB_ROWBUF;          // words of finished rows a wave holds before it stores them (a row is <= 64 words)
struct alignas(16) CBuildTask {
  int list, len;       // the list to stream: col[list .. list + len)
  unsigned off_lo;     // word offset of the task's row in the arena, low 32 bits
  unsigned off_hi_fl;  // bits 0..7 offset >> 32; 8..19 bit_off; 20..26 words of the row (1..64); 31 type B
};
struct CBuildParams {
  GraphView g;
  const ChunkRec *chunks;   // host chunks: consecutive vertices whose DAG rows fit the stage
  const int *order;         // dequeue position -> chunk id (nullptr = identity)
  int count;
  const int *trp;           // task lists: row offsets per host vertex (nv + 1)
  const CBuildTask *tasks;
  unsigned *queue;          // dequeue head (zeroed before launch)
  unsigned *mat;            // matrix arena
  int flags;
};
hipError_t launch_cbuild(const CBuildParams &p, int stage, int grid_blocks, hipStream_t stream);
int cbuild_per_cu(int stage);
// second DFS level of the narrow vertices (matrix <= kBitWords words): one workgroup per chunk of the narrow chunk table
struct CliqueSmallParams {
  const int *rp;
  const ChunkRec *chunks;
  const int *order;
  int first, step, count;            // this rank's (round's) share: chunk ids order[first + i * step]
  const unsigned long long *base;    // per vertex: word offset of its matrix in the arena (nv + 1)
  const unsigned *mat;
  unsigned *queue;
  unsigned long long *counters;      // [0] += 4-cliques
  int topo;                          // strictly upper triangular matrices: the words below j / 32 of M_j are skipped
};
hipError_t launch_clique_small(const CliqueSmallParams &p, int grid_blocks, hipStream_t stream);

// entries of a rank's support array in the several-rank diamond (gm_diamond_support_*): |E+| rounded up so that every rank's slice of the
// reduce-scatter has the same, 256-byte aligned size
__host__ __device__ inline long long diamond_support_entries(long long ne, int world) {
  const long long q = 64ll * (world > 1 ? world : 1);
  return ((ne > 0 ? ne : 1) + q - 1) / q * q;
}
// constants the byte model of bench.py needs: exported through gm_constant (gm_tools.hip) so that they cannot drift apart
constexpr int kMotifTrimMinList = 128;  // 3-motif enumeration: a partner list of >= this many keys is trimmed to its keys below max(u, v)
constexpr int kWideMinWordsDefault = kBitWords;  // k-clique: a matrix of more words is counted by the wide classes (GM_WIDE_MIN_WORDS overrides)
// k-clique: vertices of the dense hub core (gm_cgather.hip), GM_CORE_H overrides.  32 K = 128 MB of bitmap; 4-clique on R-MAT-22 ef 28 at
// 4 K .. 128 K: 32.6 / 30.6 / 29.0 / 27.8 (32 K) / 29.1 / 30.7 ms -- the gathers are bound by the lines they pull through L2
constexpr int kCoreHDefault = 32768;
constexpr int kSupStreamMaxMeanRow = 64;  // the edge supports read the key stream below this mean row length (and below 0.5 triangles per entry: gm_launch.hip)
constexpr int kTopoMinMeanRow = 0;        // DAG patterns run on the topologically renumbered copy from this mean row length (sum d+^2 / |E+|) on: always
                                          // (64 until round 6; with the hub corner and the key stream of rounds 4 - 5 the renumbered copy wins on the short-row
                                          // graphs too: TC flat 0.396 -> 0.364 ms, power law 0.674 -> 0.490, communities 0.682 -> 0.618)

// host-side launchers (gm_mine.hip)
hipError_t launch_mine(Pattern pat, const MineParams &p, int grid_blocks, hipStream_t stream);
constexpr int kTctStageMax = 2048;  // gm_tch.hip: the longest DAG row its stage takes
// the tasks of a chunk against its rows as one hashed (row, id) set in LDS: gm_tch.hip (rounds 2 - 3: a sorted LDS copy behind a bit filter,
// tct_kernel -- deleted in round 5)
hipError_t launch_tch(const MineParams &p, int stage, int grid_blocks, hipStream_t stream);
int tch_per_cu(int stage);
// edge supports t(e) = triangles through e, into p.scratch (one 32-bit counter per DAG entry, zeroed by the caller): the task lists'
// triangle pass with three increments per match (gm_sup.hip); then sum C(t, 2) over a range of entries
hipError_t launch_sup(const MineParams &p, int stage, int grid_blocks, hipStream_t stream);
int sup_per_cu(int stage);
// edge supports of the out-edges of the rows beyond the stage (gm_sup.hip sup_long_kernel)
struct SupLongParams {
  const int *rp, *col;
  const int *rows;            // the rows beyond the stage
  const long long *prefix;    // prefix[r] = task edges of the rows before r (nrows + 1)
  int nrows;
  long long total;
  unsigned *sup;
  int topo, rank, world;
};
hipError_t launch_sup_long(const SupLongParams &p, int cu_count, hipStream_t stream);
// edge supports, second pass: the match masks of the in-edge tasks summed by column into the supports (gm_sup.hip sup_cols_kernel)
struct SupColsParams {
  int nv, lmin;                  // lmin: the shortest tail that has a mask
  long long ne;
  const int *far_rows;           // the rows with tails of more than 64 keys, widest ids first (sup_far_kernel: wave w takes rows w, w + W, ...)
  int n_far_rows;
  const int *rp;
  const unsigned *emoff;         // per DAG entry: the offset of its task's mask (64-bit words), kNoMask = none
  const unsigned long long *smask;
  unsigned *sup;
};
hipError_t launch_sup_cols(const SupColsParams &p, int cu_count, hipStream_t stream);
hipError_t launch_sup_pairs(const unsigned *sup, long long first, long long count, unsigned long long *out, int cu_count, hipStream_t stream);
size_t mine_lds_bytes(Pattern pat);
// the big-LDS classes (gm_mine_wide.hip): cls = 1 (mid rows) or 2 (big rows); DIAMOND, MOTIF3, MOTIF4E only
hipError_t launch_mine_wide(Pattern pat, int cls, const MineParams &p, int grid_blocks, hipStream_t stream);
size_t mine_wide_lds_bytes(int cls);
// hashed-row classes (gm_hrow.hip): the row as a hash-partitioned set of 16-bit remainders, 2^LB buckets of eight slots
constexpr int kClassRowMin = 1024;  // rows longer than this leave the general kernel when the classes are on
#ifndef GM_HROW_LB_MID
#define GM_HROW_LB_MID 12
#endif
constexpr int kHrowLbMid = GM_HROW_LB_MID;  // class 1 (rows of 1025..8191 entries): 2^12 buckets = 64 KB (2^11: 32 KB)
constexpr int kHrowLbBig = 13;  // class 2 (rows of 8192..24576 entries): 128 KB
hipError_t launch_hrow(Pattern pat, int cls, const MineParams &p, int grid_blocks, hipStream_t stream);
size_t hrow_lds_bytes(int cls);
// giant rows (> kStageCapBig entries): pieces of kStageCapBig entries as hashed sets, chunks of kGiantEdges task edges (gm_hrow.hip)
#ifndef GM_GIANT_EDGES
#define GM_GIANT_EDGES 8192
#endif
constexpr int kGiantEdges = GM_GIANT_EDGES;
hipError_t launch_giant(Pattern pat, const MineParams &p, int grid_blocks, hipStream_t stream);
unsigned long long giant_scratch_words(int max_deg);
int giant_per_cu();
int hrow_per_cu(int cls);
// ids must split into bucket + 14-bit remainder: bits of nv <= LB_max + 14
inline bool hrow_fits(int nv, int cls) {
  int k = 0;
  while (k < 31 && (1ll << k) < (long long)nv) ++k;
  return k <= (cls == 2 ? kHrowLbBig : kHrowLbMid) + 14;
}
int mine_wide_threads(Pattern pat, int cls);

}  // namespace gm
