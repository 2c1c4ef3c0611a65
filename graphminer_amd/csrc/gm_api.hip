// gm_api.hip -- the C ABI (include/graphminer_amd.h) over the HIP kernels.
//
// Host-side counterparts in the reference:
//   GraphGPU::init / init_edgelist          include/graph_gpu.h:69-194
//   launch sizing                            src/triangle/gpu_base.cu:36-45, src/clique/gpu_base.cu:28-50
//   Scheduler::round_robin                   src/common/scheduler.cc:34-85
//   Graph::orientation                       src/common/graph.cc:233-279
// None of that code is reused: tasks are described by a compact chunk table (16 B per ~256 edges)
// instead of per-GPU COO copies, and the multi-GPU split is index arithmetic on chunk ids.
#include "../../include/graphminer_amd.h"
#include "gm_mine.h"
#include "gm_setops.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <list>
#include <mutex>
#include <string>
#include <vector>

using namespace gm;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? GM_ERR_NO_DEVICE
                                                                                                  : GM_ERR_HIP;
}
#define HIP_TRY(call)                                              \
  do {                                                             \
    hipError_t _e = (call);                                        \
    if (_e != hipSuccess) return hip_fail(_e, #call, __FILE__, __LINE__); \
  } while (0)

extern "C" const char *gm_strerror(int s) {
  switch (s) {
    case GM_OK: return "ok";
    case GM_ERR_INVALID: return "invalid argument";
    case GM_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case GM_ERR_HIP: return "HIP runtime error";
    case GM_ERR_TOO_LARGE: return "graph exceeds the 32-bit task index of this build";
    case GM_ERR_UNSUPPORTED: return "Not implemented";
    case GM_ERR_IO: return "I/O error";
    case GM_ERR_FORMAT: return "bad graph format";
    default: return "unknown status";
  }
}
extern "C" const char *gm_last_error(void) { return g_last_error.c_str(); }
extern "C" int gm_version(void) { return 100; }

extern "C" int gm_device_count(int *n) {
  if (!n) return GM_ERR_INVALID;
  *n = 0;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
  *n = c;
  return c > 0 ? GM_OK : GM_ERR_NO_DEVICE;
}

// ------------------------------------------------------------------------------------------------
// device-side setup helpers: scans / sorts / selections of the pre-processing run on the GPU (rocPRIM through hipCUB for the
// primitives, hand-written kernels around them) -- no host loop over the vertices anywhere in the default paths
// ------------------------------------------------------------------------------------------------
template <class T>
struct DevBuf {  // RAII device array
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = count;
    return hipMalloc(&p, sizeof(T) * std::max<size_t>(count, 1));
  }
  T *release() { T *q = p; p = nullptr; n = 0; return q; }
};

struct ScanTemp {  // temp storage of the hipCUB calls, grown on demand
  DevBuf<char> buf;
  hipError_t reserve(size_t bytes) { return bytes <= buf.n ? hipSuccess : buf.alloc(bytes); }
};

template <class TI, class TO>
static hipError_t dev_exclusive_sum(ScanTemp &tmp, const TI *d_in, TO *d_out, size_t n, hipStream_t stream = 0) {
  size_t bytes = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_in, d_out, (int)n, stream);
  if (e != hipSuccess) return e;
  if ((e = tmp.reserve(bytes)) != hipSuccess) return e;
  return hipcub::DeviceScan::ExclusiveSum(tmp.buf.p, bytes, d_in, d_out, (int)n, stream);
}

// ------------------------------------------------------------------------------------------------
// graph handle
// ------------------------------------------------------------------------------------------------
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
  int tct = 0;                                 // 1: chunk costs from the task lists of gm_tct.hip (the tasks a chunk hosts, not its own entries)
  __host__ __device__ bool skips(int d) const {
    return (skip_clique_wide > 0 && clique_is_wide(d, skip_clique_wide)) || (d > skip_lo && d <= skip_hi) || !(d > only_lo && d <= only_hi);
  }
  bool operator==(const RowFilter &o) const {
    return skip_clique_wide == o.skip_clique_wide && skip_lo == o.skip_lo && skip_hi == o.skip_hi && only_lo == o.only_lo && only_hi == o.only_hi && tct == o.tct;
  }
};

struct ChunkTable {
  int target;       // T: CSR entries per chunk
  bool allow_split; // rows longer than the staging capacity may be cut across chunks
  int bit_words;    // clique: LDS bit-matrix budget the chunks were built for (0 = unconstrained)
  unsigned long long part_cap = 0;  // estimated work above which a chunk is cut into parts
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
  std::vector<unsigned long long> edge_prefix;  // edges owned by chunks [0,i)
  std::vector<int> first_vertex;                // u_begin of every chunk (ascending; SPLIT chunks repeat it)
};

struct WidePlan {
  int rank = 0, world = 1, policy = 0;
  std::vector<int> verts;  // this rank's wide vertices; slot = index here (heaviest first)
  struct Round {
    size_t chunk_begin = 0, chunk_end = 0;  // row-group chunks of the round (phase 1)
    size_t cls_begin[4] = {0, 0, 0, 0};     // slots of the round per count class S / L / X, in d_cls_slots (phase 2)
    unsigned long long words = 0;           // arena words of the round
  };
  std::vector<Round> rounds;
  unsigned long long edges = 0;    // task edges of these vertices
  size_t n_chunks = 0;
  int *d_verts = nullptr;
  unsigned long long *d_base = nullptr;  // slot -> word offset inside its round's arena
  ChunkRec *d_chunks = nullptr;
  int *d_cls_slots = nullptr;
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
  int *d_col = nullptr;  // col_idx
  bool own_col = true;
  int2 *d_edesc = nullptr;  // per CSR entry: {rp[col[e]], degree(col[e])}, built on first use (ensure_edesc)
  int *d_trp = nullptr;     // task lists of the shorter-list-streams triangle count (ensure_tasklists): row offsets (nv + 1)
  int2 *d_tdesc = nullptr;  // ... and per task {rp[partner], d(partner)}
  std::vector<int> h_rp;  // host copy of the offsets, fetched on first use (host_rp): download, k-clique tables, SgL renumbering
  std::list<ChunkTable> tables;  // list: handed-out pointers stay valid
  unsigned long long *d_counters = nullptr;  // [4] + queue word, 64 B
  unsigned *d_scratch = nullptr;
  size_t scratch_bytes = 0;
  static constexpr int kEvRing = 64;  // HIP-event pairs of the most recent launches
  hipEvent_t ev[kEvRing][2] = {};
  unsigned long long ev_launches = 0;
  int cu_count = 256;
  gm_graph *dag_cache = nullptr;          // oriented copy, built on demand by gm_motif_formula
  gm_graph *relabel_cache[2] = {nullptr, nullptr};  // copies renumbered by degree (ascending / descending), see get_relabeled
  const gm_graph *ring_alias = nullptr;
  int *d_idx0 = nullptr;                  // rectangle: #neighbours below v, and the wedge-block prefix
  unsigned long long *d_wblock_prefix = nullptr;
  unsigned long long n_wblocks = 0;
  int4 *d_rect_tasks = nullptr;          // rectangle by wedge accumulation: task list, counter maps
  unsigned long long n_rect_tasks = 0;
  unsigned *d_rect_acc = nullptr;
  size_t rect_acc_bytes = 0;
  unsigned *d_house_t = nullptr;         // house by wedge accumulation: per-entry tables, task list, 64-bit maps
  unsigned *d_house_tlt = nullptr;
  int4 *d_house_tasks = nullptr;
  unsigned long long n_house_tasks = 0;
  unsigned long long *d_house_acc = nullptr;
  size_t house_acc_bytes = 0;
  int *d_house_touched = nullptr;
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
  std::list<struct WidePlan> wide_plans;
  unsigned *d_wide_mat = nullptr;      // matrix arena (largest round so far)
  size_t wide_mat_bytes = 0;
  unsigned *d_wide_queue = nullptr;    // dequeue words of the wide launches of one call (zeroed per call)
  hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};  // class kernels that cannot fill the chip run beside the others (run_pattern)
  hipEvent_t aux_done[3] = {nullptr, nullptr, nullptr};
  gm_setup_times setup = {0, 0, 0, 0, 0};  // accumulated pre-processing time of this handle (gm_graph_setup_times)
  std::mutex mu;
};

// wall-clock stopwatch for the setup accounting (host clock: the steps mix host work, copies and synchronised kernels)
struct SetupTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

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

extern "C" int gm_graph_setup_times(const gm_graph *g, gm_setup_times *out) {
  if (!g || !out) return GM_ERR_INVALID;
  *out = g->setup;
  // cached derived handles report through their owner
  for (const gm_graph *r : {g->dag_cache, g->relabel_cache[0], g->relabel_cache[1]})
    if (r) {
      out->orient_ms += r->setup.orient_ms;
      out->table_ms += r->setup.table_ms;
      out->bitmap_ms += r->setup.bitmap_ms;
      out->other_ms += r->setup.other_ms;
    }
  return GM_OK;
}

static void free_tables(gm_graph *g) {
  for (auto &t : g->tables) {
    if (t.d) (void)hipFree(t.d);
    if (t.d_slot) (void)hipFree(t.d_slot);
    if (t.d_row_slot && t.own_bitmaps) (void)hipFree(t.d_row_slot);
    for (int i = 0; i < 2; ++i) if (t.d_order[i]) (void)hipFree(t.d_order[i]);
    if (t.d_bitmaps && t.own_bitmaps) (void)hipFree(t.d_bitmaps);
  }
  g->tables.clear();
}

extern "C" void gm_graph_free(gm_graph *g) {
  if (!g) return;
  if (g->dag_cache) gm_graph_free(g->dag_cache);
  g->dag_cache = nullptr;
  for (auto &r : g->relabel_cache) {
    if (r) gm_graph_free(r);
    r = nullptr;
  }
  (void)hipSetDevice(g->device);
  free_tables(g);
  for (auto &b : g->bitmap_sets) {
    if (b.d_bitmaps) (void)hipFree(b.d_bitmaps);
    if (b.d_row_slot) (void)hipFree(b.d_row_slot);
  }
  if (g->d_rp) (void)hipFree(g->d_rp);
  if (g->own_col && g->d_col) (void)hipFree(g->d_col);
  if (g->d_edesc) (void)hipFree(g->d_edesc);
  if (g->d_trp) (void)hipFree(g->d_trp);
  if (g->d_tdesc) (void)hipFree(g->d_tdesc);
  for (auto &pl : g->wide_plans) {
    if (pl.d_verts) (void)hipFree(pl.d_verts);
    if (pl.d_base) (void)hipFree(pl.d_base);
    if (pl.d_chunks) (void)hipFree(pl.d_chunks);
    if (pl.d_cls_slots) (void)hipFree(pl.d_cls_slots);
  }
  if (g->d_wide_mat) (void)hipFree(g->d_wide_mat);
  if (g->d_wide_sorted) (void)hipFree(g->d_wide_sorted);
  if (g->d_wide_queue) (void)hipFree(g->d_wide_queue);
  for (auto &st_ : g->aux_stream) if (st_) (void)hipStreamDestroy(st_);
  for (auto &ev_ : g->aux_done) if (ev_) (void)hipEventDestroy(ev_);
  if (g->d_counters) (void)hipFree(g->d_counters);
  if (g->d_scratch) (void)hipFree(g->d_scratch);
  if (g->d_idx0) (void)hipFree(g->d_idx0);
  if (g->d_wblock_prefix) (void)hipFree(g->d_wblock_prefix);
  if (g->d_house_prefix) (void)hipFree(g->d_house_prefix);
  if (g->d_pent_touched) (void)hipFree(g->d_pent_touched);
  if (g->d_house_t) (void)hipFree(g->d_house_t);
  if (g->d_house_tlt) (void)hipFree(g->d_house_tlt);
  if (g->d_house_tasks) (void)hipFree(g->d_house_tasks);
  if (g->d_house_acc) (void)hipFree(g->d_house_acc);
  if (g->d_house_touched) (void)hipFree(g->d_house_touched);
  if (g->d_rect_tasks) (void)hipFree(g->d_rect_tasks);
  if (g->d_rect_acc) (void)hipFree(g->d_rect_acc);
  for (auto &pr : g->ev)
    for (auto &e : pr)
      if (e) (void)hipEventDestroy(e);
  delete g;
}

static int finish_handle(gm_graph *g) {
  HIP_TRY(hipMalloc(&g->d_counters, 64));
  HIP_TRY(hipMemset(g->d_counters, 0, 64));
  for (auto &pr : g->ev)
    for (auto &e : pr) HIP_TRY(hipEventCreate(&e));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g->device));
  g->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return GM_OK;  // (max_deg: set by the caller from the device-side pass that produced the offsets)
}

// host copy of the 32-bit offsets, fetched from the device on first use
static int host_rp(gm_graph *g, const std::vector<int> **out) {
  if (g->h_rp.size() != (size_t)g->nv + 1) {
    g->h_rp.resize((size_t)g->nv + 1);
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipMemcpy(g->h_rp.data(), g->d_rp, sizeof(int) * ((size_t)g->nv + 1), hipMemcpyDeviceToHost));
  }
  if (out) *out = &g->h_rp;
  return GM_OK;
}

// int64 offsets of the ABI -> the internal int32 copy, validated on the device: err bit 0 = not an offset array (first != 0,
// last != ne, or decreasing), bit 1 = a row of 2^24 entries or more; info[1] = longest row
__global__ __launch_bounds__(256) void convert_offsets_kernel(const long long *__restrict__ rp64, int nv, long long ne, int *__restrict__ rp32,
                                                              int *__restrict__ info) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  int err = 0, md = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v <= nv; v += stride) {
    const long long x = rp64[v];
    rp32[v] = (int)x;
    if (v == 0 && x != 0) err |= 1;
    if (v == nv && x != ne) err |= 1;
    if (v > 0) {
      const long long d = x - rp64[v - 1];
      if (d < 0) err |= 1;
      else if (d >= (1 << 24)) err |= 2;  // per-row limit of the flattened scan
      else md = max(md, (int)d);
    }
  }
  md = gm::wave_max_nonneg(md);
  err = gm::wave_max_nonneg(err & 1) | (gm::wave_max_nonneg((err >> 1) & 1) << 1);
  if ((threadIdx.x & 63) == 0) {
    if (md) atomicMax(&info[1], md);
    if (err & 3) atomicOr(&info[0], err & 3);
  }
}

// d_rp64: DEVICE array of nv + 1 int64 offsets. Allocates and fills g->d_rp, sets g->max_deg.
static int adopt_offsets(gm_graph *g, const int64_t *d_rp64) {
  HIP_TRY(hipMalloc(&g->d_rp, sizeof(int) * ((size_t)g->nv + 1)));
  DevBuf<int> info;
  HIP_TRY(info.alloc(2));
  HIP_TRY(hipMemset(info.p, 0, 8));
  const long long blocks = std::min<long long>(((long long)g->nv + 256) / 256, 4096);
  hipLaunchKernelGGL(convert_offsets_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, (const long long *)d_rp64, g->nv, g->ne, g->d_rp, info.p);
  int h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, info.p, 8, hipMemcpyDeviceToHost));
  if (h[0] & 1) return GM_ERR_FORMAT;
  if (h[0] & 2) return GM_ERR_TOO_LARGE;
  g->max_deg = h[1];
  return GM_OK;
}

static int check_sizes(long long nv, long long ne) {
  if (nv < 0 || ne < 0) return GM_ERR_INVALID;
  if (nv >= 0x7ffffffeLL || ne >= 0x7fffffffLL) return GM_ERR_TOO_LARGE;
  return GM_OK;
}

static int convert_offsets(const int64_t *rp64, int nv, long long ne, std::vector<int> &out) {
  out.resize((size_t)nv + 1);
  if (rp64[0] != 0 || rp64[nv] != ne) return GM_ERR_FORMAT;
  for (int v = 0; v <= nv; ++v) {
    if (v > 0 && rp64[v] < rp64[v - 1]) return GM_ERR_FORMAT;
    if (v > 0 && rp64[v] - rp64[v - 1] >= (1 << 24)) return GM_ERR_TOO_LARGE;  // per-row limit of the flattened scan
    out[v] = (int)rp64[v];
  }
  return GM_OK;
}

extern "C" int gm_graph_upload(const gm_csr *h, int device, gm_graph **out) {
  if (!h || !out || !h->row_ptr || (h->ne > 0 && !h->col_idx)) return GM_ERR_INVALID;
  *out = nullptr;
  int rc = check_sizes(h->nv, h->ne);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  gm_graph *g = new gm_graph();
  g->device = device;
  g->nv = h->nv;
  g->ne = h->ne;
  auto fail = [&](int code) { gm_graph_free(g); return code; };
  {  // the int64 offsets go up as they are and are narrowed / validated on the device (adopt_offsets)
    DevBuf<int64_t> rp64;
    hipError_t e;
    if ((e = rp64.alloc((size_t)g->nv + 1)) != hipSuccess) return fail(hip_fail(e, "hipMalloc(rp64)", __FILE__, __LINE__));
    if ((e = hipMemcpy(rp64.p, h->row_ptr, sizeof(int64_t) * ((size_t)g->nv + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy(rp)", __FILE__, __LINE__));
    rc = adopt_offsets(g, rp64.p);
    if (rc) return fail(rc);
  }
  hipError_t e;
  if ((e = hipMalloc(&g->d_col, sizeof(int) * (size_t)std::max<long long>(g->ne, 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc(col)", __FILE__, __LINE__));
  if (g->ne > 0 && (e = hipMemcpy(g->d_col, h->col_idx, sizeof(int) * (size_t)g->ne, hipMemcpyHostToDevice)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy(col)", __FILE__, __LINE__));
  rc = finish_handle(g);
  if (rc) return fail(rc);
  *out = g;
  return GM_OK;
}

extern "C" int gm_graph_from_device(int32_t nv, int64_t ne, const int64_t *d_row_ptr, const int32_t *d_col_idx, int device,
                                    gm_graph **out) {
  if (!out || !d_row_ptr || (ne > 0 && !d_col_idx)) return GM_ERR_INVALID;
  *out = nullptr;
  int rc = check_sizes(nv, ne);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(device));
  gm_graph *g = new gm_graph();
  g->device = device;
  g->nv = nv;
  g->ne = ne;
  g->own_col = false;
  g->d_col = const_cast<int *>(d_col_idx);
  auto fail = [&](int code) { gm_graph_free(g); return code; };
  rc = adopt_offsets(g, d_row_ptr);
  if (rc) return fail(rc);
  rc = finish_handle(g);
  if (rc) return fail(rc);
  *out = g;
  return GM_OK;
}

// Graph::sort_neighbors (src/common/graph.cc:138-146: std::sort per row under OpenMP) on the GPU: one segmented radix sort
// of col_idx with the rows as segments. In place: a borrowed col_idx array (gm_graph_from_device) is overwritten too.
extern "C" int gm_graph_sort_neighbors(gm_graph *g) {
  if (!g) return GM_ERR_INVALID;
  if (g->ne == 0) return GM_OK;
  if (!g->tables.empty() || g->d_edesc || g->dag_cache || g->relabel_cache[0] || g->relabel_cache[1]) return GM_ERR_INVALID;  // before any solver ran
  HIP_TRY(hipSetDevice(g->device));
  DevBuf<int> sorted;
  HIP_TRY(sorted.alloc((size_t)g->ne));
  ScanTemp tmp;
  size_t bytes = 0;
  int bits = 1;
  while (bits < 31 && (1ll << bits) < (long long)g->nv) ++bits;
  HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, bytes, g->d_col, sorted.p, (int)g->ne, g->nv, g->d_rp, g->d_rp + 1, 0, bits));
  HIP_TRY(tmp.reserve(bytes));
  HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(tmp.buf.p, bytes, g->d_col, sorted.p, (int)g->ne, g->nv, g->d_rp, g->d_rp + 1, 0, bits));
  HIP_TRY(hipMemcpy(g->d_col, sorted.p, sizeof(int) * (size_t)g->ne, hipMemcpyDeviceToDevice));
  HIP_TRY(hipDeviceSynchronize());
  return GM_OK;
}

extern "C" int gm_graph_meta(const gm_graph *g, gm_csr *m) {
  if (!g || !m) return GM_ERR_INVALID;
  m->nv = g->nv;
  m->ne = g->ne;
  m->max_deg = g->max_deg;
  m->row_ptr = nullptr;
  m->col_idx = nullptr;
  return GM_OK;
}

extern "C" int gm_graph_download(const gm_graph *g, int64_t *row_ptr, int32_t *col_idx) {
  if (!g) return GM_ERR_INVALID;
  if (row_ptr) {
    const std::vector<int> *rp = nullptr;
    int rc = host_rp(const_cast<gm_graph *>(g), &rp);
    if (rc) return rc;
    for (int v = 0; v <= g->nv; ++v) row_ptr[v] = (*rp)[(size_t)v];
  }
  if (col_idx && g->ne > 0) {
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipMemcpy(col_idx, g->d_col, sizeof(int) * (size_t)g->ne, hipMemcpyDeviceToHost));
  }
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// orientation on the GPU (Graph::orientation, src/common/graph.cc:233-279)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dag_keep(int ds, int s, int dd, int d) { return dd > ds || (dd == ds && d > s); }

// Orientation kernels: count (pass 0) or compact (pass 1) the kept neighbours, order preserved (ballot + popcount
// ranks). Short rows (<= kOrientShort entries) take 8 lanes each, 8 rows per wave. Longer rows are cut into
// SEGMENTS of kOrientSeg entries (table built on the host from the row offsets) so that a 100k-entry hub row is
// spread over ~100 waves instead of serialising one.
constexpr int kOrientShort = 64;
constexpr int kOrientSeg = 1024;

struct OrientSeg {
  int row, begin, end, out;  // CSR entry range [begin,end) of `row`; out = output offset of the segment (pass 1)
};

__global__ __launch_bounds__(256) void orient_short_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ col,
                                                           int *__restrict__ new_deg, const int *__restrict__ new_rp,
                                                           int *__restrict__ new_col, int pass) {
  constexpr int G = 8, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G, gl = lane % G;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = wave * RPW; s0 < nv; s0 += nwaves * RPW) {
    const int s = s0 + grp;
    int b = 0, ds = 0;
    if (s < nv) { b = rp[s]; ds = rp[s + 1] - b; }
    const int full = ds;
    if (ds > kOrientShort) ds = 0;  // long rows belong to the segment kernel
    const int maxds = wave_max_nonneg(ds);
    int n = 0;
    const int ob = (pass && ds > 0) ? new_rp[s] : 0;
    for (int base = 0; base < maxds; base += G) {
      const int i = base + gl;
      bool keep = false;
      int d = 0;
      if (i < ds) {
        d = col[b + i];
        keep = dag_keep(full, s, rp[d + 1] - rp[d], d);
      }
      const unsigned long long m = (__ballot(keep) >> (grp * G)) & 0xffull;
      if (pass && keep) new_col[ob + n + __popcll(m & ((1ull << gl) - 1ull))] = d;
      n += __popcll(m);
    }
    if (!pass && gl == 0 && s < nv && full <= kOrientShort) new_deg[s] = n;
  }
}

// one wave per segment; pass 0 adds the segment's count to its row's new degree (and keeps it per segment for the offsets)
__global__ __launch_bounds__(256) void orient_seg_kernel(int nseg, const OrientSeg *__restrict__ segs, const int *__restrict__ rp,
                                                         const int *__restrict__ col, int *__restrict__ seg_count, int *__restrict__ new_deg,
                                                         int *__restrict__ new_col, int pass) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int sg = wave; sg < nseg; sg += nwaves) {
    const OrientSeg q = segs[sg];
    const int ds = rp[q.row + 1] - rp[q.row];
    int n = 0;
    for (int base = q.begin; base < q.end; base += 64) {
      const int i = base + lane;
      bool keep = false;
      int d = 0;
      if (i < q.end) {
        d = col[i];
        keep = dag_keep(ds, q.row, rp[d + 1] - rp[d], d);
      }
      const unsigned long long m = __ballot(keep);
      if (pass && keep) new_col[q.out + n + rank_below(m)] = d;
      n += __popcll(m);
    }
    if (!pass && lane == 0) {
      seg_count[sg] = n;
      if (n) atomicAdd(&new_deg[q.row], n);
    }
  }
}

// segment table of the long rows, built on the device: seg_first[v] = exclusive scan of ceil(d/kOrientSeg) over the long rows
__global__ __launch_bounds__(256) void orient_segcount_kernel(int nv, const int *__restrict__ rp, int *__restrict__ nseg_of) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nv) return;
  int n = 0;
  if (v < nv) {
    const int d = rp[v + 1] - rp[v];
    n = d > kOrientShort ? (d + kOrientSeg - 1) / kOrientSeg : 0;
  }
  nseg_of[v] = n;  // (nseg_of[nv] = 0: the scan's last element is the total)
}
__global__ __launch_bounds__(256) void orient_segfill_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ seg_first,
                                                             OrientSeg *__restrict__ segs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const int b = rp[v], e = rp[v + 1];
  if (e - b <= kOrientShort) return;
  int k = seg_first[v];
  for (int s0 = b; s0 < e; s0 += kOrientSeg) segs[k++] = {v, s0, min(s0 + kOrientSeg, e), 0};
}
// output offset of every segment: the row's new offset + the kept entries of the row's earlier segments (one thread per long row)
__global__ __launch_bounds__(256) void orient_segout_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ seg_first,
                                                            const int *__restrict__ seg_count, const int *__restrict__ new_rp,
                                                            OrientSeg *__restrict__ segs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  if (rp[v + 1] - rp[v] <= kOrientShort) return;
  int run = new_rp[v];
  for (int k = seg_first[v]; k < seg_first[v + 1]; ++k) {
    segs[k].out = run;
    run += seg_count[k];
  }
}
__global__ __launch_bounds__(256) void max_degree_kernel(int nv, const int *__restrict__ rp, int *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  int md = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) md = max(md, rp[v + 1] - rp[v]);
  md = gm::wave_max_nonneg(md);
  if ((threadIdx.x & 63) == 0 && md) atomicMax(out, md);
}

extern "C" int gm_graph_orient(const gm_graph *sym, gm_graph **out) {
  if (!sym || !out) return GM_ERR_INVALID;
  *out = nullptr;
  HIP_TRY(hipSetDevice(sym->device));
  SetupTimer timer;
  const int nv = sym->nv;
  const unsigned vb = (unsigned)((nv + 256) / 256);  // blocks covering v = 0 .. nv
  ScanTemp tmp;
  // segment table of the long rows (device): counts -> exclusive scan -> fill
  DevBuf<int> nseg_of, seg_first, deg, segcnt;
  DevBuf<OrientSeg> segs;
  HIP_TRY(nseg_of.alloc((size_t)nv + 1));
  HIP_TRY(seg_first.alloc((size_t)nv + 1));
  hipLaunchKernelGGL(orient_segcount_kernel, dim3(vb), dim3(256), 0, 0, nv, sym->d_rp, nseg_of.p);
  HIP_TRY(dev_exclusive_sum(tmp, nseg_of.p, seg_first.p, (size_t)nv + 1));
  int nseg = 0;
  HIP_TRY(hipMemcpy(&nseg, seg_first.p + nv, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(segs.alloc((size_t)nseg));
  HIP_TRY(segcnt.alloc((size_t)nseg));
  if (nseg) hipLaunchKernelGGL(orient_segfill_kernel, dim3(vb), dim3(256), 0, 0, nv, sym->d_rp, seg_first.p, segs.p);
  // pass 0: new degrees (short rows write, segments of long rows add)
  HIP_TRY(deg.alloc((size_t)nv + 1));
  HIP_TRY(hipMemsetAsync(deg.p, 0, sizeof(int) * ((size_t)nv + 1), 0));
  const int bs = std::max(1, std::min((nv + 31) / 32, sym->cu_count * 8));
  const int bl = std::max(1, std::min((nseg + 3) / 4, sym->cu_count * 8));
  hipLaunchKernelGGL(orient_short_kernel, dim3(bs), dim3(256), 0, 0, nv, sym->d_rp, sym->d_col, deg.p, (const int *)nullptr, (int *)nullptr, 0);
  if (nseg)
    hipLaunchKernelGGL(orient_seg_kernel, dim3(bl), dim3(256), 0, 0, nseg, segs.p, sym->d_rp, sym->d_col, segcnt.p, deg.p, (int *)nullptr, 0);
  // new offsets = exclusive scan of the new degrees (parallel_prefix_sum, include/scan.h:5-35)
  gm_graph *g = new gm_graph();
  g->device = sym->device;
  g->nv = nv;
  auto fail = [&](int code) { gm_graph_free(g); return code; };
  hipError_t e;
  if ((e = hipMalloc(&g->d_rp, sizeof(int) * ((size_t)nv + 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  if ((e = dev_exclusive_sum(tmp, deg.p, g->d_rp, (size_t)nv + 1)) != hipSuccess) return fail(hip_fail(e, "ExclusiveSum", __FILE__, __LINE__));
  DevBuf<int> md;
  if ((e = md.alloc(1)) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  (void)hipMemsetAsync(md.p, 0, sizeof(int), 0);
  hipLaunchKernelGGL(max_degree_kernel, dim3((unsigned)std::min<long long>(((long long)nv + 255) / 256, 2048)), dim3(256), 0, 0, nv, g->d_rp, md.p);
  int ne_new = 0, max_deg = 0;
  if ((e = hipMemcpy(&ne_new, g->d_rp + nv, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy", __FILE__, __LINE__));
  if ((e = hipMemcpy(&max_deg, md.p, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy", __FILE__, __LINE__));
  g->ne = ne_new;
  g->max_deg = max_deg;
  if ((e = hipMalloc(&g->d_col, sizeof(int) * (size_t)std::max(ne_new, 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  // pass 1: compact
  if (nseg) hipLaunchKernelGGL(orient_segout_kernel, dim3(vb), dim3(256), 0, 0, nv, sym->d_rp, seg_first.p, segcnt.p, g->d_rp, segs.p);
  hipLaunchKernelGGL(orient_short_kernel, dim3(bs), dim3(256), 0, 0, nv, sym->d_rp, sym->d_col, (int *)nullptr, g->d_rp, g->d_col, 1);
  if (nseg)
    hipLaunchKernelGGL(orient_seg_kernel, dim3(bl), dim3(256), 0, 0, nseg, segs.p, sym->d_rp, sym->d_col, (int *)nullptr, (int *)nullptr, g->d_col, 1);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(hip_fail(e, "orient kernels", __FILE__, __LINE__));
  int rc = finish_handle(g);
  if (rc) { gm_graph_free(g); return rc; }
  g->setup.orient_ms = timer.ms();
  *out = g;
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// Degree renumbering. A pattern count does not depend on the vertex numbering, but the work of the SgL kernels does: they
// anchor a match at its largest vertex id and walk smaller ids. With ids ascending in degree the 2-path walks of rectangle and the
// (v0, v1 < v0, v3) tasks of house go through low-degree vertices (R-MAT-16: 522 M -> 128 M 2-paths, 898 M -> 163 M tasks); with
// ids descending in degree the wedges (v0; v2 < v1 < v0) of pentagon avoid the hubs (99 M -> 33 M). The copy is built once
// per handle: counting sort of the vertices by degree on the host, one 64-bit key (new row, new neighbour) per CSR entry and
// a device radix sort (hipCUB) -- the rows come out ascending.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relabel_keys_kernel(int nv, long long ne, const int *__restrict__ rp, const int *__restrict__ col,
                                                           const int *__restrict__ newid, unsigned long long *__restrict__ keys) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  int lo = 0, hi = nv - 1;  // row of entry e
  while (lo < hi) {
    const int mid = (int)(((long long)lo + hi + 1) >> 1);
    if (rp[mid] <= e) lo = mid; else hi = mid - 1;
  }
  keys[e] = ((unsigned long long)(unsigned)newid[lo] << 32) | (unsigned long long)(unsigned)newid[col[e]];
}

__global__ __launch_bounds__(256) void relabel_cols_kernel(long long ne, const unsigned long long *__restrict__ keys, int *__restrict__ col) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) col[e] = (int)(unsigned)(keys[e] & 0xffffffffull);
}

static int get_relabeled(gm_graph *g, int descending, gm_graph **out) {
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->relabel_cache[descending]) { *out = g->relabel_cache[descending]; return GM_OK; }
  }
  HIP_TRY(hipSetDevice(g->device));
  SetupTimer timer;
  {
    int rc = host_rp(g, nullptr);
    if (rc) return rc;
  }
  const int nv = g->nv;
  const long long ne = g->ne;
  // counting sort by degree (ties: ascending id)
  std::vector<int> newid((size_t)std::max(nv, 1));
  {
    std::vector<long long> bucket((size_t)g->max_deg + 2, 0);
    for (int v = 0; v < nv; ++v) bucket[(size_t)(g->h_rp[v + 1] - g->h_rp[v]) + 1]++;
    for (size_t d = 1; d < bucket.size(); ++d) bucket[d] += bucket[d - 1];
    for (int v = 0; v < nv; ++v) {
      const long long pos = bucket[(size_t)(g->h_rp[v + 1] - g->h_rp[v])]++;
      newid[(size_t)v] = descending ? (int)((long long)nv - 1 - pos) : (int)pos;
    }
  }
  gm_graph *r = new gm_graph();
  r->device = g->device;
  r->nv = nv;
  r->ne = ne;
  r->h_rp.assign((size_t)nv + 1, 0);
  for (int v = 0; v < nv; ++v) r->h_rp[(size_t)newid[(size_t)v] + 1] = g->h_rp[v + 1] - g->h_rp[v];
  for (int v = 0; v < nv; ++v) r->h_rp[(size_t)v + 1] += r->h_rp[(size_t)v];
  int *d_newid = nullptr;
  unsigned long long *d_keys = nullptr, *d_sorted = nullptr;
  void *d_tmp = nullptr;
  auto cleanup = [&]() {
    if (d_newid) (void)hipFree(d_newid);
    if (d_keys) (void)hipFree(d_keys);
    if (d_sorted) (void)hipFree(d_sorted);
    if (d_tmp) (void)hipFree(d_tmp);
  };
  auto fail = [&](hipError_t e, const char *what) { cleanup(); gm_graph_free(r); return hip_fail(e, what, __FILE__, __LINE__); };
  hipError_t e;
  const size_t n1 = (size_t)std::max<long long>(ne, 1);
  if ((e = hipMalloc(&d_newid, sizeof(int) * (size_t)std::max(nv, 1))) != hipSuccess) return fail(e, "hipMalloc(newid)");
  if ((e = hipMemcpy(d_newid, newid.data(), sizeof(int) * (size_t)nv, hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(newid)");
  if ((e = hipMalloc(&d_keys, sizeof(unsigned long long) * n1)) != hipSuccess) return fail(e, "hipMalloc(keys)");
  if ((e = hipMalloc(&d_sorted, sizeof(unsigned long long) * n1)) != hipSuccess) return fail(e, "hipMalloc(sorted)");
  if ((e = hipMalloc(&r->d_rp, sizeof(int) * ((size_t)nv + 1))) != hipSuccess) return fail(e, "hipMalloc(rp)");
  if ((e = hipMalloc(&r->d_col, sizeof(int) * n1)) != hipSuccess) return fail(e, "hipMalloc(col)");
  if ((e = hipMemcpy(r->d_rp, r->h_rp.data(), sizeof(int) * ((size_t)nv + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(rp)");
  if (ne > 0) {
    const unsigned blocks = (unsigned)((ne + 255) / 256);
    hipLaunchKernelGGL(relabel_keys_kernel, dim3(blocks), dim3(256), 0, 0, nv, ne, g->d_rp, g->d_col, d_newid, d_keys);
    int bits = 1;
    while (bits < 32 && (1ll << bits) < (long long)nv) ++bits;
    size_t tmp_bytes = 0;
    if ((e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_keys, d_sorted, (int)ne, 0, 32 + bits)) != hipSuccess) return fail(e, "SortKeys(size)");
    if ((e = hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 16))) != hipSuccess) return fail(e, "hipMalloc(sort temp)");
    if ((e = hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_bytes, d_keys, d_sorted, (int)ne, 0, 32 + bits)) != hipSuccess) return fail(e, "SortKeys");
    hipLaunchKernelGGL(relabel_cols_kernel, dim3(blocks), dim3(256), 0, 0, ne, d_sorted, r->d_col);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "relabel kernels");
  }
  cleanup();
  int rc = finish_handle(r);
  if (rc) { gm_graph_free(r); return rc; }
  r->max_deg = g->max_deg;  // (a permutation of the same rows)
  std::lock_guard<std::mutex> lk(g->mu);
  g->relabel_cache[descending] = r;
  g->setup.relabel_ms += timer.ms();
  *out = r;
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// task chunk tables
// ------------------------------------------------------------------------------------------------
// The greedy chunk walk over the vertices [v0, v1): a contiguous run of whole rows is closed when it reaches `target` entries,
// would exceed the LDS stage (or, k-clique, the bit-matrix budget), spans kMaxChunkVerts rows, or meets a row that is left out /
// too long for the stage; rows longer than the stage are SPLIT into `target`-entry chunks (allow_split) or chunks of their
// own. Runs on the host (gm_chunk_table, GM_HOST_TABLES) and, one thread per block of kTableBlock vertices, on the device:
// the walk restarts at every block boundary, so a table is the same whichever side built it.
constexpr int kTableBlock = 2048;
struct ChunkWalk {
  int target, allow_split, bit_words, stage_cap;
  RowFilter rf;
};
template <class Emit>
__host__ __device__ inline unsigned long long walk_chunks(const ChunkWalk &w, const int *rp, int v0, int v1, Emit emit) {
  unsigned long long max_bit_words = 0;
  int u = v0;
  while (u < v1) {
    const int d = rp[u + 1] - rp[u];
    if (d == 0) { ++u; continue; }
    if (w.rf.skips(d)) { ++u; continue; }
    if (d > w.stage_cap) {
      if (w.allow_split) {
        for (int s0 = rp[u]; s0 < rp[u + 1]; s0 += w.target) emit(ChunkRec{u, u + 1, s0, min(s0 + w.target, rp[u + 1]), 0, 1, GM_WAVE, 0});
      } else {
        emit(ChunkRec{u, u + 1, rp[u], rp[u + 1], 0, 1, GM_WAVE, 0});
        if (w.bit_words) max_bit_words = max(max_bit_words, (unsigned long long)d * (unsigned long long)((d + 31) / 32));
      }
      ++u;
      continue;
    }
    const int start = u;
    int edges = 0, maxd = 0;
    while (u < v1 && (u - start) < kMaxChunkVerts) {
      const int du = rp[u + 1] - rp[u];
      if (du > w.stage_cap) break;
      if (w.rf.skips(du) && du > 0) break;
      if (edges > 0 && edges + du > w.stage_cap) break;
      if (w.bit_words && edges > 0) {
        const int nm = max(maxd, du);
        if ((long long)(edges + du) * ((nm + 31) / 32) > w.bit_words) break;
      }
      edges += du;
      maxd = max(maxd, du);
      ++u;
      if (edges >= w.target) break;
    }
    if (edges > 0) {
      emit(ChunkRec{start, u, rp[start], rp[u], 0, 1, GM_WAVE, 0});
      if (w.bit_words) {
        const unsigned long long bw = (unsigned long long)edges * (unsigned long long)((maxd + 31) / 32);
        if (bw > (unsigned long long)w.bit_words) max_bit_words = max(max_bit_words, bw);
      }
    } else if (u == start) {
      ++u;  // (cannot happen: the row at `start` fits the stage and is not filtered)
    }
  }
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
                                                         const int2 *__restrict__ tdesc = nullptr) {
  const ChunkRec r = chunks[blockIdx.x];
  unsigned long long c = 0;
  if (owner_rule == 2) {  // gm_tct.hip: the keys of the lists this chunk's vertices host
    for (int te = trp[r.u_begin] + (int)threadIdx.x; te < trp[r.u_end]; te += 256) c += (unsigned long long)tdesc[te].y + 8ull;
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
// A workgroup (one wave) owns kWalkPerWG blocks: all 64 lanes copy the blocks' offsets into LDS with coalesced loads (128 KB of
// the CU's 160 KB), then lanes 0 .. kWalkPerWG-1 each walk one block out of LDS -- the walk is a serial chain of dependent
// reads, ~30 cycles per vertex from LDS against a memory round trip per cache line from HBM (one thread per block straight
// from memory: 132 ms for the three 3-motif tables of R-MAT-24; this form: see profiles/r02/ab_setup_device_tables.log).
constexpr int kWalkPerWG = 16;
__global__ __launch_bounds__(64) void tab_walk_kernel(ChunkWalk w, int nv, const int *__restrict__ rp, int nblocks, int *__restrict__ count,
                                                      unsigned long long *__restrict__ max_bw, const int *__restrict__ offset, ChunkRec *__restrict__ recs) {
  __shared__ int rpl[kWalkPerWG][kTableBlock + 1];  // (row stride 2049 words: the walkers' lanes fall into different banks)
  const int b0 = blockIdx.x * kWalkPerWG;
  for (int k = 0; k < kWalkPerWG; ++k) {
    const int v0 = (b0 + k) * kTableBlock;
    if (v0 >= nv) break;
    const int n = min(kTableBlock, nv - v0) + 1;
    for (int i = threadIdx.x; i < n; i += 64) rpl[k][i] = rp[v0 + i];
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
    const unsigned long long want = (cost[c] + cap - 1) / cap;
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
__global__ __launch_bounds__(256) void tab_orderkeys_kernel(int n, const unsigned long long *__restrict__ cost, unsigned long long heavy, int classes_only,
                                                            unsigned long long *__restrict__ key0, unsigned long long *__restrict__ key1, int *__restrict__ iota) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const unsigned long long x = cost[c];
  key0[c] = x >= heavy ? (classes_only ? 1ull : x) : 0ull;
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

static int build_table_device(gm_graph *g, ChunkTable &t, bool sym_table, double &bitmap_ms) {
  const int nv = g->nv;
  TableDevParams q;
  q.nv = nv;
  q.rf = t.rf;
  ScanTemp tmp;
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  // the greedy walk, one thread per block of kTableBlock vertices: count, scan, emit
  ChunkWalk w{t.target, t.allow_split ? 1 : 0, t.bit_words, t.stage_cap, t.rf};
  const int nblk = (nv + kTableBlock - 1) / kTableBlock;
  DevBuf<int> bcount, boff;
  DevBuf<unsigned long long> maxbw;
  HIP_TRY(bcount.alloc((size_t)nblk + 1));
  HIP_TRY(boff.alloc((size_t)nblk + 1));
  HIP_TRY(maxbw.alloc(1));
  HIP_TRY(hipMemsetAsync(maxbw.p, 0, 8, 0));
  const dim3 wgrid((unsigned)((nblk + 1 + kWalkPerWG - 1) / kWalkPerWG));
  hipLaunchKernelGGL(tab_walk_kernel, wgrid, dim3(64), 0, 0, w, nv, g->d_rp, nblk, bcount.p, maxbw.p, (const int *)nullptr, (ChunkRec *)nullptr);
  HIP_TRY(dev_exclusive_sum(tmp, bcount.p, boff.p, (size_t)nblk + 1));
  int n0 = 0;
  HIP_TRY(hipMemcpy(&n0, boff.p + nblk, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&t.max_bit_words, maxbw.p, 8, hipMemcpyDeviceToHost));
  t.n = 0;
  HIP_TRY(hipMalloc(&t.d, sizeof(ChunkRec)));  // (placeholder, replaced below when the table has chunks)
  if (n0 == 0) {
    t.edge_prefix.assign(1, 0ull);
    return GM_OK;
  }
  DevBuf<ChunkRec> recs0;
  HIP_TRY(recs0.alloc((size_t)n0));
  hipLaunchKernelGGL(tab_walk_kernel, wgrid, dim3(64), 0, 0, w, nv, g->d_rp, nblk, bcount.p, maxbw.p, (const int *)boff.p, recs0.p);
  // estimated work per chunk, parts
  DevBuf<unsigned long long> cost0;
  HIP_TRY(cost0.alloc((size_t)n0));
  HIP_TRY(hipMemsetAsync(cost0.p, 0, sizeof(unsigned long long) * (size_t)n0, 0));
  hipLaunchKernelGGL(chunk_cost_kernel, dim3((unsigned)n0), dim3(256), 0, 0, g->d_rp, g->d_col, recs0.p, cost0.p, t.rf.tct ? 2 : (sym_table ? 1 : 0), kStageCapWide,
                     g->d_trp, g->d_tdesc);
  DevBuf<int> np, bsz, off;
  HIP_TRY(np.alloc((size_t)n0 + 1));
  HIP_TRY(bsz.alloc((size_t)n0 + 1));
  HIP_TRY(off.alloc((size_t)n0 + 1));
  hipLaunchKernelGGL(tab_parts_kernel, blocks(n0 + 1), dim3(256), 0, 0, n0, recs0.p, g->d_rp, cost0.p, t.stage_cap,
                     std::max<unsigned long long>(t.part_cap, 1), (t.allow_split || sym_table) ? 1 : 0, np.p, bsz.p);
  HIP_TRY(dev_exclusive_sum(tmp, np.p, off.p, (size_t)n0 + 1));
  int n = 0;
  HIP_TRY(hipMemcpy(&n, off.p + n0, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(t.d);
  t.d = nullptr;
  HIP_TRY(hipMalloc(&t.d, sizeof(ChunkRec) * (size_t)n));
  DevBuf<unsigned long long> cost;
  DevBuf<int> edges, firstv;
  HIP_TRY(cost.alloc((size_t)n));
  HIP_TRY(edges.alloc((size_t)n));
  HIP_TRY(firstv.alloc((size_t)n));
  hipLaunchKernelGGL(tab_expand_kernel, blocks(n0), dim3(256), 0, 0, n0, recs0.p, cost0.p, np.p, bsz.p, off.p, t.d, cost.p, edges.p, firstv.p);
  t.n = (size_t)n;
  // host views of the per-chunk scalars (O(chunks), not O(vertices)): edges -> prefix, first vertex, cost
  {
    std::vector<int> h_edges((size_t)n);
    t.first_vertex.resize((size_t)n);
    t.cost.resize((size_t)n);
    HIP_TRY(hipMemcpy(h_edges.data(), edges.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t.first_vertex.data(), firstv.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t.cost.data(), cost.p, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    t.edge_prefix.resize((size_t)n + 1);
    t.edge_prefix[0] = 0;
    for (size_t i = 0; i < (size_t)n; ++i) t.edge_prefix[i + 1] = t.edge_prefix[i] + (unsigned long long)h_edges[i];
  }
  // dequeue orders: stable descending radix sorts of (key, chunk id)
  {
    unsigned long long total_cost = 0;
    for (auto c : t.cost) total_cost += c;
    const unsigned long long heavy = 2ull * (total_cost / (unsigned long long)n) + 1ull;
    DevBuf<unsigned long long> key0, key1, keyo;
    DevBuf<int> iota;
    HIP_TRY(key0.alloc((size_t)n));
    HIP_TRY(key1.alloc((size_t)n));
    HIP_TRY(keyo.alloc((size_t)n));
    HIP_TRY(iota.alloc((size_t)n));
    hipLaunchKernelGGL(tab_orderkeys_kernel, blocks(n), dim3(256), 0, 0, n, cost.p, heavy, sym_table ? 1 : 0, key0.p, key1.p, iota.p);
    for (int m = 0; m < 2; ++m) {
      HIP_TRY(hipMalloc(&t.d_order[m], sizeof(int) * (size_t)n));
      size_t bytes = 0;
      const unsigned long long *keys = m == 0 ? key0.p : key1.p;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, keys, keyo.p, iota.p, t.d_order[m], n));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp.buf.p, bytes, keys, keyo.p, iota.p, t.d_order[m], n));
      t.order[m].resize((size_t)n);
      HIP_TRY(hipMemcpy(t.order[m].data(), t.d_order[m], sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    }
  }
  if (t.allow_split) {
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
        HIP_TRY(hipMemcpy(hv.data(), sel.p, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hd.data(), degs.p, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost));
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
        HIP_TRY(hipMemcpy(d_rows.p, rows.data(), sizeof(int) * nb, hipMemcpyHostToDevice));
        HIP_TRY(row_slot.alloc((size_t)nv));
        HIP_TRY(hipMemsetAsync(row_slot.p, 0xff, sizeof(int) * (size_t)nv, 0));  // -1
        hipLaunchKernelGGL(tab_rowslot_kernel, blocks((long long)nb), dim3(256), 0, 0, (int)nb, d_rows.p, row_slot.p);
        HIP_TRY(bitmaps.alloc((size_t)nb * (size_t)words));
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
      HIP_TRY(hipMalloc(&t.d_slot, sizeof(int) * (size_t)n));
      hipLaunchKernelGGL(tab_chunkslot_kernel, blocks(n), dim3(256), 0, 0, n, t.d, g->d_rp, t.stage_cap, t.d_row_slot, t.d_slot);
    }
    bitmap_ms = bm_timer.ms();
  }
  HIP_TRY(hipDeviceSynchronize());
  return GM_OK;
}

static int get_table(gm_graph *g, int target, bool allow_split, int bit_words, unsigned long long part_cap, int stage_cap,
                     ChunkTable **out, const RowFilter &rf = RowFilter(), int bitmap_min_deg = kBitmapMinDeg) {
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
  if (!getenv("GM_HOST_TABLES")) {  // (GM_HOST_TABLES: the same walk in a host loop over the vertices, kept for A/B)
    HIP_TRY(hipSetDevice(g->device));
    int rc = build_table_device(g, t, sym_table, bitmap_ms);
    if (rc) return rc;
    if (getenv("GM_TABLE_INFO")) {
      unsigned long long tc = 0, mx = 0;
      for (auto c : t.cost) { tc += c; mx = std::max(mx, c); }
      fprintf(stderr, "[table/device] stage_cap %d rows (%d,%d] skip (%d,%d]: %zu chunks, est. keys %.3e (max chunk %.3e), %zu bitmaps, edges %llu, %.2f ms\n",
              stage_cap, rf.only_lo, rf.only_hi, rf.skip_lo, rf.skip_hi, t.n, (double)tc, (double)mx, t.n_bitmaps, t.edge_prefix.back(), timer.ms());
    }
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
    hipError_t e = hipMalloc(&d_tmp, sizeof(ChunkRec) * recs.size());
    if (e == hipSuccess) e = hipMalloc(&d_cost, sizeof(unsigned long long) * recs.size());
    if (e == hipSuccess) e = hipMemcpy(d_tmp, recs.data(), sizeof(ChunkRec) * recs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cost, 0, sizeof(unsigned long long) * recs.size());
    if (e == hipSuccess) {
      hipLaunchKernelGGL(chunk_cost_kernel, dim3((unsigned)recs.size()), dim3(256), 0, 0, g->d_rp, g->d_col, d_tmp, d_cost,
                         sym_table ? 1 : 0, kStageCapWide);
      e = hipMemcpy(cost.data(), d_cost, sizeof(unsigned long long) * recs.size(), hipMemcpyDeviceToHost);
    }
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_cost) (void)hipFree(d_cost);
    if (e != hipSuccess) return hip_fail(e, "chunk_cost_kernel", __FILE__, __LINE__);
  }
  if (allow_split || sym_table) {  // (clique chunks are never cut: their second phase needs the whole bit-matrix)
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
      const int np = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)std::max(batches / min_batches, 1), (cost[i] + cap - 1) / cap));
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
  HIP_TRY(hipMalloc(&t.d, sizeof(ChunkRec) * std::max<size_t>(t.n, 1)));
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
      HIP_TRY(hipMalloc(&t.d_order[m], sizeof(int) * t.n));
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
      HIP_TRY(hipMalloc(&t.d_slot, sizeof(int) * t.n));
      HIP_TRY(hipMemcpy(t.d_slot, slots.data(), sizeof(int) * t.n, hipMemcpyHostToDevice));
      {
        std::vector<int> by_vertex((size_t)g->nv, -1);
        for (size_t i = 0; i < nb; ++i) by_vertex[(size_t)rows[i]] = (int)i;
        HIP_TRY(hipMalloc(&t.d_row_slot, sizeof(int) * (size_t)g->nv));
        HIP_TRY(hipMemcpy(t.d_row_slot, by_vertex.data(), sizeof(int) * (size_t)g->nv, hipMemcpyHostToDevice));
      }
      HIP_TRY(hipMalloc(&t.d_bitmaps, (size_t)nb * (size_t)words * 4));
      HIP_TRY(hipMemset(t.d_bitmaps, 0, (size_t)nb * (size_t)words * 4));
      int *d_rows = nullptr;
      HIP_TRY(hipMalloc(&d_rows, sizeof(int) * nb));
      HIP_TRY(hipMemcpy(d_rows, rows.data(), sizeof(int) * nb, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(bitmap_build_kernel, dim3((unsigned)nb), dim3(256), 0, 0, g->d_rp, g->d_col, d_rows, t.d_bitmaps, words);
      hipError_t e = hipDeviceSynchronize();
      (void)hipFree(d_rows);
      if (e != hipSuccess) return hip_fail(e, "bitmap_build_kernel", __FILE__, __LINE__);
      t.n_bitmaps = nb;
      t.bitmap_words = words;
    }
    bitmap_ms = bm_timer.ms();
  }
  HIP_TRY(hipDeviceSynchronize());
  if (getenv("GM_TABLE_INFO")) {  // diagnostics
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

// ---- task lists of gm_tct.hip: every edge u -> v of the DAG is a task of the endpoint with the longer out-list -------------------
__global__ __launch_bounds__(256) void task_keys_kernel(int nv, long long ne, const int *__restrict__ rp, const int *__restrict__ col,
                                                         unsigned long long *__restrict__ keys, int *__restrict__ cnt, int stage_max) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) {
    int lo = 0, hi = nv - 1;  // the row of entry e: largest u with rp[u] <= e
    while (lo < hi) {
      const int mid = (int)(((long long)lo + hi + 1) >> 1);
      if (rp[mid] <= e) lo = mid; else hi = mid - 1;
    }
    const int u = lo, v = col[e];
    const int du = rp[u + 1] - rp[u], dv = rp[v + 1] - rp[v];
    if (du > stage_max) {  // a row the stage cannot take hosts nothing: its out-edges stay with the chunked kernel (run_pattern)
      keys[e] = ~0ull;     // (sorts behind every task)
      continue;
    }
    const bool u_hosts = dv > stage_max || du >= dv;  // the longer list hosts (ties: the source) -- unless it does not fit the stage
    const int host = u_hosts ? u : v, partner = u_hosts ? v : u;
    keys[e] = ((unsigned long long)(unsigned)host << 32) | (unsigned)partner;
    atomicAdd(&cnt[host], 1);
  }
}
__global__ __launch_bounds__(256) void task_desc_kernel(long long ne, const int *__restrict__ rp, const unsigned long long *__restrict__ sorted,
                                                         int2 *__restrict__ tdesc) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) {
    if (sorted[e] == ~0ull) { tdesc[e] = make_int2(0, 0); continue; }  // (the out-edges of rows beyond the stage: not tasks)
    const int y = (int)(unsigned)(sorted[e] & 0xffffffffull);
    const int r = rp[y];
    tdesc[e] = make_int2(r, rp[y + 1] - r);
  }
}
static int ensure_tasklists(gm_graph *g) {
  if (g->d_tdesc || g->ne == 0) return GM_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->d_tdesc) return GM_OK;
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  const size_t ne = (size_t)g->ne, nv1 = (size_t)g->nv + 1;
  DevBuf<unsigned long long> keys, sorted;
  DevBuf<int> cnt;
  ScanTemp tmp;
  HIP_TRY(keys.alloc(ne));
  HIP_TRY(sorted.alloc(ne));
  HIP_TRY(cnt.alloc(nv1));
  HIP_TRY(hipMemset(cnt.p, 0, sizeof(int) * nv1));
  const long long blocks = std::min<long long>(((long long)ne + 255) / 256, (long long)g->cu_count * 32);
  hipLaunchKernelGGL(task_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->nv, g->ne, g->d_rp, g->d_col, keys.p, cnt.p, kTctStageMax);
  int bits = 1;
  while (bits < 32 && (1ll << bits) < (long long)g->nv) ++bits;
  size_t bytes = 0;
  const int end_bit = g->max_deg > kTctStageMax ? 64 : 32 + bits;  // (the all-ones keys of excluded edges need every bit)
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, keys.p, sorted.p, (int)ne, 0, end_bit));
  HIP_TRY(tmp.reserve(bytes));
  HIP_TRY(hipcub::DeviceRadixSort::SortKeys(tmp.buf.p, bytes, keys.p, sorted.p, (int)ne, 0, end_bit));
  int *trp = nullptr;
  int2 *td = nullptr;
  HIP_TRY(hipMalloc(&trp, sizeof(int) * nv1));
  hipError_t e = dev_exclusive_sum(tmp, cnt.p, trp, nv1);
  if (e == hipSuccess) e = hipMalloc(&td, sizeof(int2) * ne);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(task_desc_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->ne, g->d_rp, sorted.p, td);
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) { (void)hipFree(trp); if (td) (void)hipFree(td); return hip_fail(e, "task lists", __FILE__, __LINE__); }
  g->d_trp = trp;
  g->d_tdesc = td;
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

static int ensure_edesc(gm_graph *g) {
  if (g->d_edesc || g->ne == 0) return GM_OK;
  std::lock_guard<std::mutex> lk(g->mu);
  if (g->d_edesc) return GM_OK;
  SetupTimer timer;
  int2 *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(int2) * (size_t)g->ne));
  const long long blocks = std::min<long long>((g->ne + 255) / 256, (long long)g->cu_count * 32);
  hipLaunchKernelGGL(edesc_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->ne, g->d_rp, g->d_col, d);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e, "edesc_kernel", __FILE__, __LINE__); }
  g->d_edesc = d;
  g->setup.table_ms += timer.ms();
  return GM_OK;
}

// k-clique, wide vertices (see gm_mine.h): the plan of one rank's share -- which wide vertices it owns (every world-th of
// the list sorted by row length, or a contiguous range), where each one's matrix sits in the arena, the row-group chunks of
// phase 1 and the slots per count class of phase 2. The arena is bounded (GM_WIDE_ARENA_MB, default 16 GiB): a share whose
// matrices need more is processed in several ROUNDS that reuse it.
#ifndef GM_WIDE_MIN_WORDS_DEFAULT
#define GM_WIDE_MIN_WORDS_DEFAULT kBitWords
#endif
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
                                                         const int *__restrict__ rp, int *__restrict__ verts, int *__restrict__ degs, int *__restrict__ ngroups) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > count) return;
  if (i == count) { ngroups[i] = 0; return; }
  const int u = wide_sorted[first + (long long)i * step];
  const int d = rp[u + 1] - rp[u];
  verts[i] = u;
  degs[i] = d;
  const int R = clique_group_rows(d);
  ngroups[i] = (d + R - 1) / R;
}
__global__ __launch_bounds__(256) void wide_groups_kernel(int count, const int *__restrict__ verts, const int *__restrict__ rp, const int *__restrict__ goff,
                                                          int batch, ChunkRec *__restrict__ chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int u = verts[i], b = rp[u], d = rp[u + 1] - b, R = clique_group_rows(d);
  int o = goff[i];
  for (int g0 = 0; g0 < d; g0 += R) chunks[o++] = {u, u + 1, b + g0, b + min(g0 + R, d), 0, 1, batch, i + 1};
}

static int clique_wide_min_words() {
  static const int v = [] {
    const char *e = getenv("GM_WIDE_MIN_WORDS");  // (sweeps; read once: tables and plans are cached per graph)
    return e ? std::max(64, std::min(atoi(e), kBitWords)) : GM_WIDE_MIN_WORDS_DEFAULT;
  }();
  return v;
}

static int get_wide_plan(gm_graph *g, int rank, int world, int policy, WidePlan **out) {
  std::lock_guard<std::mutex> lk(g->mu);
  SetupTimer timer;
  HIP_TRY(hipSetDevice(g->device));
  ScanTemp tmp;
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  if (!g->wide_valid) {  // once per graph: the wide vertices, longest rows first (select + stable radix sort by row length)
    const int nv = g->nv;
    DevBuf<int> flag, iota, sel, nsel, degs, keyo;
    HIP_TRY(flag.alloc((size_t)nv));
    HIP_TRY(iota.alloc((size_t)nv));
    HIP_TRY(sel.alloc((size_t)nv));
    HIP_TRY(nsel.alloc(1));
    hipLaunchKernelGGL(wide_flag_kernel, blocks(nv), dim3(256), 0, 0, nv, g->d_rp, clique_wide_min_words(), flag.p, iota.p);
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
    HIP_TRY(tmp.reserve(bytes));
    HIP_TRY(hipcub::DeviceSelect::Flagged(tmp.buf.p, bytes, iota.p, flag.p, sel.p, nsel.p, nv));
    int m = 0;
    HIP_TRY(hipMemcpy(&m, nsel.p, sizeof(int), hipMemcpyDeviceToHost));
    g->n_wide = (size_t)m;
    if (m > 0) {
      HIP_TRY(degs.alloc((size_t)m));
      HIP_TRY(keyo.alloc((size_t)m));
      hipLaunchKernelGGL(gather_deg_kernel, blocks(m), dim3(256), 0, 0, m, sel.p, g->d_rp, degs.p);
      HIP_TRY(hipMalloc(&g->d_wide_sorted, sizeof(int) * (size_t)m));
      bytes = 0;
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, degs.p, keyo.p, sel.p, g->d_wide_sorted, m));
      HIP_TRY(tmp.reserve(bytes));
      HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(tmp.buf.p, bytes, degs.p, keyo.p, sel.p, g->d_wide_sorted, m));
    }
    g->wide_valid = true;
  }
  for (auto &pl : g->wide_plans)
    if (pl.rank == rank && pl.world == world && pl.policy == policy) { *out = &pl; return GM_OK; }
  WidePlan pl;
  pl.rank = rank; pl.world = world; pl.policy = policy;
  int64_t first = 0, step = 1, count = 0;
  gm_partition((int64_t)g->n_wide, rank, world, policy, &first, &step, &count);
  if (count > 0) {
    DevBuf<int> degs, ngroups, goff;
    HIP_TRY(hipMalloc(&pl.d_verts, sizeof(int) * (size_t)count));
    HIP_TRY(degs.alloc((size_t)count));
    HIP_TRY(ngroups.alloc((size_t)count + 1));
    HIP_TRY(goff.alloc((size_t)count + 1));
    hipLaunchKernelGGL(wide_share_kernel, blocks(count + 1), dim3(256), 0, 0, (int)count, (long long)first, (long long)step, g->d_wide_sorted, g->d_rp,
                       pl.d_verts, degs.p, ngroups.p);
    HIP_TRY(dev_exclusive_sum(tmp, ngroups.p, goff.p, (size_t)count + 1));
    int nchunks = 0;
    HIP_TRY(hipMemcpy(&nchunks, goff.p + count, sizeof(int), hipMemcpyDeviceToHost));
    pl.n_chunks = (size_t)nchunks;
    HIP_TRY(hipMalloc(&pl.d_chunks, sizeof(ChunkRec) * (size_t)std::max(nchunks, 1)));
    const int build_batch = kBuildBatchRows;
    hipLaunchKernelGGL(wide_groups_kernel, blocks(count), dim3(256), 0, 0, (int)count, pl.d_verts, g->d_rp, goff.p, build_batch, pl.d_chunks);
    // host part, O(wide vertices of this share): arena offsets, rounds within the arena budget, count classes
    pl.verts.resize((size_t)count);
    std::vector<int> hd((size_t)count), hgoff((size_t)count + 1);
    HIP_TRY(hipMemcpy(pl.verts.data(), pl.d_verts, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hd.data(), degs.p, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hgoff.data(), goff.p, sizeof(int) * ((size_t)count + 1), hipMemcpyDeviceToHost));
    unsigned long long arena_mb = GM_WIDE_ARENA_MB;
    if (const char *e = getenv("GM_WIDE_ARENA_MB")) arena_mb = std::max(1ll, atoll(e));  // (tests: force several rounds)
    const unsigned long long budget_words = (arena_mb << 20) / 4ull;
    std::vector<unsigned long long> base((size_t)count);
    std::vector<int> cls_slots;
    size_t s0 = 0;
    while (s0 < (size_t)count) {
      WidePlan::Round rd;
      rd.chunk_begin = (size_t)hgoff[s0];
      size_t s1 = s0;
      unsigned long long words = 0;
      std::vector<int> by_cls[3];
      for (; s1 < (size_t)count; ++s1) {
        const int d = hd[s1];
        const unsigned long long w = (unsigned long long)d * (unsigned long long)((d + 31) / 32);
        if (s1 > s0 && words + w > budget_words) break;
        base[s1] = words;
        words += w;
        pl.edges += (unsigned long long)d;
        by_cls[clique_count_class(d)].push_back((int)s1);
      }
      rd.chunk_end = (size_t)hgoff[s1];
      rd.words = words;
      for (int c = 0; c < 3; ++c) {
        rd.cls_begin[c] = cls_slots.size();
        cls_slots.insert(cls_slots.end(), by_cls[c].begin(), by_cls[c].end());
      }
      rd.cls_begin[3] = cls_slots.size();
      pl.rounds.push_back(rd);
      s0 = s1;
    }
    HIP_TRY(hipMalloc(&pl.d_base, sizeof(unsigned long long) * base.size()));
    HIP_TRY(hipMemcpy(pl.d_base, base.data(), sizeof(unsigned long long) * base.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&pl.d_cls_slots, sizeof(int) * cls_slots.size()));
    HIP_TRY(hipMemcpy(pl.d_cls_slots, cls_slots.data(), sizeof(int) * cls_slots.size(), hipMemcpyHostToDevice));
    unsigned long long need_words = 0;
    for (auto &rd : pl.rounds) need_words = std::max(need_words, rd.words);
    const size_t need = (size_t)need_words * 4;
    if (need > g->wide_mat_bytes) {
      if (g->d_wide_mat) (void)hipFree(g->d_wide_mat);
      g->d_wide_mat = nullptr;
      g->wide_mat_bytes = 0;
      HIP_TRY(hipMalloc(&g->d_wide_mat, need));
      g->wide_mat_bytes = need;
    }
    if (!g->d_wide_queue) HIP_TRY(hipMalloc(&g->d_wide_queue, 65536));
    HIP_TRY(hipDeviceSynchronize());
  }
  g->wide_plans.push_back(std::move(pl));
  *out = &g->wide_plans.back();
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

// ------------------------------------------------------------------------------------------------
// solvers
// ------------------------------------------------------------------------------------------------
enum FinMode : int { FIN_COPY = 0, FIN_MOTIF3 = 1, FIN_MOTIF3_FORMULA = 2, FIN_RAW4 = 3, FIN_HALF_SIGNED = 4 };

__global__ void finalize_kernel(int mode, unsigned long long base, const unsigned long long *__restrict__ c,
                                unsigned long long *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (mode == FIN_MOTIF3) {
    out[0] = c[2] - c[0];  // wedges = sum_e idx(e) - sum_e |A' ^ B|   (automine_base.h:13)
    out[1] = c[1];         // triangles                                  (automine_base.h:18)
  } else if (mode == FIN_MOTIF3_FORMULA) {
    out[0] = base - 3ull * c[0];  // wedges = sum_v C(d,2) - 3T  (src/motif/omp_formula.cc:39-40); base only on rank 0
    out[1] = c[0];
  } else if (mode == FIN_RAW4) {
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  } else if (mode == FIN_HALF_SIGNED) {
    out[0] = (unsigned long long)((long long)c[0] >> 1);  // an even two's-complement sum (pent_acc_kernel): rank partials add up mod 2^64
  } else {
    out[0] = c[0];
  }
}

// ---- common launch prologue / epilogue of every mining entry point -----------------------------------------------
struct LaunchCtx {
  gm_graph *g = nullptr;
  gm_launch la;          // caller's launch descriptor, or the all-zero default
  int world = 1, rank = 0;
  hipStream_t stream = nullptr;
  hipEvent_t *evp = nullptr;  // event pair of this launch (ring slot)
};

// validates the arguments, selects the device, zeroes the 64-byte counter block on the launch stream
static int begin_launch(const gm_graph *cg, const gm_launch *la, const uint64_t *h_out, LaunchCtx &c) {
  if (!cg) return GM_ERR_INVALID;
  c.g = const_cast<gm_graph *>(cg);
  memset(&c.la, 0, sizeof c.la);
  if (la) c.la = *la;
  c.world = c.la.world > 1 ? c.la.world : 1;
  c.rank = c.la.rank;
  if (c.rank < 0 || c.rank >= c.world) return GM_ERR_INVALID;
  if (!h_out && !c.la.d_counts) return GM_ERR_INVALID;
  HIP_TRY(hipSetDevice(c.g->device));
  c.stream = (hipStream_t)c.la.stream;
  HIP_TRY(hipMemsetAsync(c.g->d_counters, 0, 64, c.stream));
  return GM_OK;
}

static int start_timer(LaunchCtx &c) {
  c.g->ring_alias = nullptr;
  c.evp = c.g->ev[c.g->ev_launches % gm_graph::kEvRing];
  c.g->ev_launches++;
  HIP_TRY(hipEventRecord(c.evp[0], c.stream));
  return GM_OK;
}

// stops the timer, publishes the counters: to d_counts (device, asynchronous) and / or to h_out (synchronises)
static int end_launch(LaunchCtx &c, int fin_mode, unsigned long long fin_base, uint64_t *h_out, int nout, gm_stats *st) {
  HIP_TRY(hipEventRecord(c.evp[1], c.stream));
  if (c.la.d_counts) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, c.stream, fin_mode, fin_base, c.g->d_counters,
                       (unsigned long long *)c.la.d_counts);
    HIP_TRY(hipGetLastError());
    if (!h_out) return GM_OK;  // asynchronous: the caller owns the synchronisation
  }
  unsigned long long v[4];
  HIP_TRY(hipMemcpyAsync(v, c.g->d_counters, sizeof v, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  if (st) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c.evp[0], c.evp[1]));
    st->kernel_ms = ms;
  }
  if (fin_mode == FIN_MOTIF3) {
    if (nout > 0) h_out[0] = v[2] - v[0];
    if (nout > 1) h_out[1] = v[1];
  } else if (fin_mode == FIN_MOTIF3_FORMULA) {
    if (nout > 0) h_out[0] = fin_base - 3ull * v[0];
    if (nout > 1) h_out[1] = v[0];
  } else if (fin_mode == FIN_RAW4) {
    for (int i = 0; i < 4 && i < nout; ++i) h_out[i] = v[i];
  } else if (fin_mode == FIN_HALF_SIGNED) {
    h_out[0] = (uint64_t)((long long)v[0] >> 1);
  } else {
    h_out[0] = v[0];
  }
  return GM_OK;
}

static void fill_stats(gm_stats *st, uint64_t tasks, uint64_t chunks, int grid, int block) {
  if (!st) return;
  st->kernel_ms = 0.0;
  st->tasks = tasks;
  st->chunks = chunks;
  st->grid = (uint32_t)grid;
  st->block = (uint32_t)block;
}

static int run_pattern(Pattern pat, const gm_graph *cg, const gm_launch *la, int k, uint64_t *h_out, int nout, gm_stats *st,
                       int fin_mode = -1, unsigned long long fin_base = 0) {
  if (fin_mode < 0) fin_mode = (pat == PAT_MOTIF3) ? FIN_MOTIF3 : FIN_COPY;
  LaunchCtx ctx;
  int rc0 = begin_launch(cg, la, h_out, ctx);
  if (rc0) return rc0;
  gm_graph *g = ctx.g;
  la = &ctx.la;
  const int world = ctx.world, rank = ctx.rank;
  hipStream_t stream = ctx.stream;

  // tune[0] = chunk target override, tune[1] = grab, tune[2] = cost_x_step, tune[3] = cost_y_step,
  // tune[4] = blocks per CU override, tune[5] = force "search in HBM" (no LDS staging) when 1
  // default chunk size: as large as the LDS stage allows (fewer dequeues, better staging reuse) while every rank still
  // gets >= ~2 chunks per resident workgroup for the dynamic dequeue to balance (matters for strong scaling at N = 8)
  int target = kDefaultChunk;
  while (target > 128 && g->ne / ((long long)world * target) < 2LL * g->cu_count * 7) target >>= 1;
  if (la->chunk > 0) target = la->chunk;
  if (la->tune[0] > 0) target = la->tune[0];
  target = std::max(64, std::min(target, kStageCap));
  const bool clique = pat == PAT_CLIQUE4 || pat == PAT_CLIQUEK;
  ChunkTable *tab = nullptr;
  // tune[6] & 0x1000 (tests): cut every chunk above 4096 estimated entries into parts
  // chunk costs are estimated keys (DAG patterns: d(u) + d(v) per edge; symmetric patterns: streamed keys, bitmap probes
  // weighted kProbeCost); parts bound the longest task of a launch
  // TC: the shorter list of every edge is streamed against the longer one (gm_tct.hip) when every DAG row fits the LDS stage
  // (tune[6] & 0x4000000: A/B switch, the chunked kernel that streams N+(v) of every out-edge).  Its chunks host the tasks of
  // their vertices -- a hub hosts 10^5 in-edges -- so their cost is counted from the task lists and heavy chunks are cut into
  // parts of 1 M keys / world (>= 128 K): one-GPU simulation of an 8-rank share of R-MAT-22, parts of 8 M / 512 K / 128 K / 32 K keys:
  // 5.36 / 1.18 / 0.99 / 1.15 ms per rank (one GPU: 6.50 / 6.54 / 6.73 / 8.09 ms), profiles/r02/ab_tct_part_cap.log
  const bool use_tct = pat == PAT_TC && !(la->tune[6] & 0x4000000) && la->tune[5] != 1 && g->ne > 0 && !getenv("GM_HOST_TABLES");
  const int tct_stage = g->max_deg <= kStageCap ? kStageCap : kTctStageMax;
  // (rows beyond the 2048-entry stage host nothing: their out-edges are the tasks of the chunked kernel, on a table of those rows only)
  const bool tct_long = use_tct && g->max_deg > kTctStageMax;
  const unsigned long long part_cap = (la->tune[6] & 0x1000) ? 4096ull : (stage_cap_of(pat) != kStageCapWide
       ? (use_tct ? std::max<unsigned long long>((1ull << 20) / (unsigned long long)world, 128ull << 10) : kPartCostCap)
       // a rank's share is 1/world of the launch: so is the tolerable tail (3-motif's bounded lists make its estimates
       // pessimistic already: measured, 1/8 share 84.8 ms unscaled vs 91.2 ms scaled; diamond 5.2 vs 4.2 ms)
       : (pat == PAT_MOTIF3 ? kPartCostCapSym : std::max<unsigned long long>(kPartCostCapSym / (unsigned long long)world, 256ull << 10)));
  // 4-clique: vertices whose matrix exceeds the 8 KB budget go through the two-phase path (gm_mine.h)
  // (tune[6] & 0x40000: A/B switch, everything stays in the mining kernel with its arena path)
  const bool use_wide = pat == PAT_CLIQUE4 && !(la->tune[6] & 0x40000);
  // symmetric-graph patterns: the rows of more than kClassRowMin entries go to the workgroup classes (gm_hrow.hip: hashed sets in LDS)
  const bool sym_pat = stage_cap_of(pat) == kStageCapWide;
  // (tune[6] & 0x80000: A/B switch, every row through the general kernel -- SPLIT chunks and dense HBM bitmaps for the long ones)
  bool use_classes = sym_pat && !(la->tune[6] & 0x80000) && !(la->tune[5] == 1);
  // They are separate launches: one whose share of chunks cannot fill the chip runs on a side stream (below), and with that the
  // classes win or tie wherever there are long rows (profiles/r02/ab_class_threshold.log, general path vs classes, ms: diamond R-MAT-16
  // 1.68 vs 0.76, R-MAT-18 2.64 vs 2.03, power law 3.94 vs 3.68, R-MAT-22 27.9 vs 16.6, R-MAT-23 255 vs 107; 3-motif R-MAT-16 1.43 vs 0.75,
  // R-MAT-20 6.4 vs 5.7, power law 4.07 vs 4.10, R-MAT-24 458 vs 172). A graph without such rows skips their (empty) tables.
  // tune[6] & 0x100000 forces them on.
  if (use_classes && !(la->tune[6] & 0x100000)) use_classes = g->max_deg > kClassRowMin;
  if (use_tct) {
    int rc_t = ensure_tasklists(g);
    if (rc_t) return rc_t;
  }
  RowFilter rf;
  rf.tct = use_tct ? 1 : 0;
  if (tct_long) { rf.skip_lo = kTctStageMax; rf.skip_hi = 0x7fffffff; }
  rf.skip_clique_wide = use_wide ? clique_wide_min_words() : 0;
  // Rows of 1025..3072 entries fit the general kernel's stage, but their partner lists (mean 600 keys on R-MAT-24) are cheaper
  // against a hashed set than against the filter + bisection of a multi-row chunk: measured (profiles/r02/ab_hrow_class_lower_bound.log,
  // ms, lower bound 3072 / 2048 / 1024 / 512 / 256) diamond R-MAT-24 262 / 240 / 239 / 238 / 237, R-MAT-22 21.1 / 21.0 / 17.5 / 17.6 / 17.5,
  // 3-motif R-MAT-24 185 / 172 / 172 / 171 / 171.
  int cls_lo = kClassRowMin;
  if (const char *e = getenv("GM_CLS_LO")) cls_lo = std::max(64, atoi(e));  // (sweeps)
  // giant rows (> kStageCapBig entries): hashed sets of row pieces (giant_kernel, gm_hrow.hip) instead of SPLIT chunks probing
  // dense bitmaps in HBM (tune[6] & 0x1000000: A/B switch, they stay SPLIT chunks of the general kernel)
  const bool use_range = use_classes && !(la->tune[6] & 0x1000000) && hrow_fits(g->nv, 2);  // (its pieces are class-2 sets: nv <= 2^27)
  if (use_classes) { rf.skip_lo = cls_lo; rf.skip_hi = use_range ? 0x7fffffff : kStageCapBig; }
  int rc = get_table(g, target, !clique, clique ? kBitWords : 0, part_cap, use_tct ? tct_stage : stage_cap_of(pat), &tab, rf, use_classes ? kStageCapBig : kBitmapMinDeg);
  if (rc) return rc;
  ChunkTable *tab_long = nullptr;
  if (tct_long) {
    RowFilter rl;
    rl.only_lo = kTctStageMax;
    rc = get_table(g, target, true, 0, kPartCostCap, kStageCap, &tab_long, rl, kBitmapMinDeg);
    if (rc) return rc;
  }
  ChunkTable *tab_cls[4] = {tab, nullptr, nullptr, nullptr};
  if (use_classes) {
    RowFilter r1, r2;
    r1.only_lo = cls_lo; r1.only_hi = kStageCapMid;
    r2.only_lo = kStageCapMid; r2.only_hi = kStageCapBig;
    // Parts of the one-row chunks: coarse. Measured on MI355X (profiles/r02/ab_sym_classes.log, diamond / 3-motif R-MAT-24, ms):
    // whole rows (largest chunk 6e7 estimated keys) 636 / 370; parts of 512 K keys 1277 / 765 (16-wave workgroups with ~10
    // batches per part); 512 K keys with >= 64 batches per part and 4-edge batches 743 / 462 -- every batch pays a few
    // dependent global round trips (descriptors, the bounded prefix of 3-motif) before it streams, so small batches and
    // small parts lose more than the shorter tail wins (8 M keys 368 / 634, 32 M keys 363 / 627). Parts of 32 M keys only trim the
    // few heaviest rows. Side streams for the class kernels: 396 / 667 (GM_CLASSES_STREAMS, off).
    // (hashed sets: a part costs one table build, ~10 us. A rank's share is 1/world of the launch and so is the tolerable tail -- one-GPU
    // simulation of 8 rank shares, parts of 32 M / 8 M / 2 M keys: 3-motif R-MAT-24 28.2 / 25.8 / 26.5 ms per rank, diamond R-MAT-22 8.3 / 6.4 / 5.4;
    // on one GPU the same parts cost 171 / 173 / 181 and 17.5 / 18.3 / 18.4 ms: profiles/r02/ab_class_part_cap.log)
    unsigned long long cls_cap = (la->tune[6] & 0x1000) ? 4096ull
                                 : std::max<unsigned long long>(part_cap, std::max<unsigned long long>((32ull << 20) / (unsigned long long)std::max(world, 1), 2ull << 20));
    if (const char *e = getenv("GM_CLS_CAP_MKEYS")) cls_cap = (unsigned long long)std::max(1, atoi(e)) << 20;  // (sweeps)
    // (target 1: every row is a chunk of its own -- the class kernels take one-row chunks -- also below the general kernel's chunk target)
    rc = get_table(g, 1, false, 0, cls_cap, kStageCapMid, &tab_cls[1], r1, 0x7fffffff);
    if (rc) return rc;
    rc = get_table(g, 1, false, 0, cls_cap, kStageCapBig, &tab_cls[2], r2, 0x7fffffff);
    if (rc) return rc;
    if (use_range) {  // pieces of kGiantEdges task edges, never cut into parts
      RowFilter r3;
      r3.only_lo = kStageCapBig;
      rc = get_table(g, kGiantEdges, true, 0, ~0ull, kStageCapBig, &tab_cls[3], r3, 0x7fffffff);
      if (rc) return rc;
    }
  }
  WidePlan *plan = nullptr;
  if (use_wide) {
    rc = get_wide_plan(g, rank, world, la->policy == GM_PART_VERTEX ? GM_PART_RANGE : la->policy, &plan);
    if (rc) return rc;
  }

  MineParams p;
  memset(&p, 0, sizeof p);
  p.g.nv = g->nv;
  p.g.ne = (int)g->ne;
  p.g.rp = g->d_rp;
  p.g.col = g->d_col;
  rc = ensure_edesc(g);  // (the kernels read them unconditionally; -DGM_EDESC=0 builds gather rp[v] instead, for A/B runs)
  if (rc) return rc;
  p.g.edesc = g->d_edesc;
  if (use_tct) {
    p.g.trp = g->d_trp;
    p.g.tdesc = g->d_tdesc;
  }
  unsigned long long my_edges = 0;
  // this rank's share of a table: chunk ids first + i*step of the dequeue order (or a contiguous / vertex range)
  auto take_share = [&](ChunkTable *tb, MineParams &q) {
    q.chunks = tb->d;
    q.chunk_slot = tb->d_slot;
    q.bitmaps = tb->d_bitmaps;
    q.bitmap_words = tb->bitmap_words;
    q.row_slot = tb->d_row_slot;
    const long long n = (long long)tb->n;
    long long first = 0, step = 1, count = 0;
    if (la->policy == GM_PART_VERTEX) {  // contiguous chunk range whose first vertex lies in this rank's vertex range
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
      gm_partition((int64_t)n, rank, world, la->policy, (int64_t *)&first, (int64_t *)&step, (int64_t *)&count);
    }
    q.first = (int)first;
    q.step = (int)step;
    q.count = (int)count;
    // dequeue order (tune[6] & 0x4000: plain chunk-id order; & 0x2000: swap the two orders -- ablation only)
    // measured on R-MAT (one rank): the cliques want the full cost order (4-clique 220.6 -> 208.6 ms, 5-clique 796 -> 589 ms),
    // the symmetric-graph patterns the locality-preserving heavy-first order at every world size, TC heavy-first for one
    // rank and the full order for shares
    const bool clique_pat = pat == PAT_CLIQUE4 || pat == PAT_CLIQUEK;
    const int which = (((world > 1 && !sym_pat) || clique_pat) ? 1 : 0) ^ ((la->tune[6] & 0x2000) ? 1 : 0);
    const bool lpt = la->policy == GM_PART_ROUND_ROBIN && tb->d_order[which] && !(la->tune[6] & 0x4000);
    q.order = lpt ? tb->d_order[which] : nullptr;
    if (step == 1) my_edges += tb->edge_prefix[first + count] - tb->edge_prefix[first];  // (any order: the same set)
    else for (long long j = first; j < n; j += step) {
      const size_t c = lpt ? (size_t)tb->order[which][(size_t)j] : (size_t)j;
      my_edges += tb->edge_prefix[c + 1] - tb->edge_prefix[c];
    }
  };
  take_share(tab, p);
  p.grab = la->tune[1] > 0 ? la->tune[1] : 1;
  // direction rule: X if b*(xb + xs*lg a) <= a*(yb + ys*lg b); tune[2] = xs+1, tune[3] = ys+1, tune[7] = xb*16 + yb
  p.cost_x_step = la->tune[2] > 0 ? la->tune[2] - 1 : 1;
  p.cost_y_step = la->tune[3] > 0 ? la->tune[3] - 1 : 6;
  p.cost_x_base = (la->tune[7] & 255) > 0 ? ((la->tune[7] >> 4) & 15) : 2;
  p.cost_y_base = (la->tune[7] & 255) > 0 ? (la->tune[7] & 15) : 2;
  p.cost_y_bitmap = la->tune[7] >> 8;  // 0 = price pass Y as a bisection even when row v has a bitmap
  p.k = k;
  p.flags = (la->tune[5] == 1) ? 1 : 0;
  p.flags |= (la->tune[6] & 0xffff) << 1;  // debug/ablation: bit1 skip clique phase 2, bit2 skip bit-matrix writes (counts wrong)
  if (la->tune[6] & 0x2000000) p.flags |= 1 << 23;  // hashed-row classes: the 32-bit multiply of id spaces beyond 2^24 (tests)
  if (la->tune[6] & 0x800000) p.flags |= 1 << 22;  // hashed-row classes: every lookup through the global-memory fallback (tests)
  if (la->tune[6] & 0x200000) p.flags |= 1 << 20;  // k >= 5: the any-width pair count instead of the tile walk (tests)
  p.counters = g->d_counters;
  p.queue = reinterpret_cast<unsigned *>(g->d_counters + 4);

  const size_t lds = mine_lds_bytes(pat);
  int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / lds));
  if (la->tune[4] > 0) per_cu = la->tune[4];
  long long want = ((long long)p.count + (long long)p.grab - 1) / (long long)p.grab;
  int grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)g->cu_count * per_cu));

  if (clique && tab->max_bit_words > 0) {
    // one arena slot per workgroup; k >= 5 doubles it (second half: compacted sub-matrices, cliquek_count_sub)
    // k >= 5: k - 2 slots of max_bit_words + 4096 words (the vertex's matrix + one compacted sub-matrix per deeper level; the
    // margin covers the padding of compacted rows to 64 columns)
    const unsigned long long slot_words = (pat == PAT_CLIQUEK) ? (unsigned long long)(k - 2) * (tab->max_bit_words + 4096ull) : tab->max_bit_words;
    const size_t need = (size_t)slot_words * sizeof(unsigned) * (size_t)grid;
    if (need > g->scratch_bytes) {
      if (g->d_scratch) (void)hipFree(g->d_scratch);
      g->d_scratch = nullptr;
      g->scratch_bytes = 0;
      HIP_TRY(hipMalloc(&g->d_scratch, need));
      g->scratch_bytes = need;
    }
    p.scratch = g->d_scratch;
    p.scratch_words = slot_words;
  }

#ifdef GM_DEBUG_CHUNKS
  unsigned long long *d_ticks = nullptr;
  HIP_TRY(hipMalloc(&d_ticks, sizeof(unsigned long long) * std::max<size_t>(tab->n, 1)));
  HIP_TRY(hipMemset(d_ticks, 0, sizeof(unsigned long long) * std::max<size_t>(tab->n, 1)));
  p.chunk_ticks = d_ticks;
#endif
  if (use_classes) {
    // everything the class launches may allocate or create, before the timer starts (a first call used to time hipMalloc and
    // hipStreamCreate between its two events): side streams + their events, and the giant-row kernel's scratch -- per workgroup,
    // where the pieces of the row cut the partner lists of its chunk, kGiantEdges ints per piece (giant_bounds)
    for (int i = 0; i < 3; ++i) {
      if (!g->aux_stream[i]) {
        HIP_TRY(hipStreamCreateWithFlags(&g->aux_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g->aux_done[i], hipEventDisableTiming));
      }
    }
    if (tab_cls[3] && tab_cls[3]->n > 0) {
      const size_t need = (size_t)giant_scratch_words(g->max_deg) * sizeof(unsigned) * (size_t)g->cu_count * (size_t)giant_per_cu();
      if (need > g->scratch_bytes) {
        if (g->d_scratch) (void)hipFree(g->d_scratch);
        g->d_scratch = nullptr;
        g->scratch_bytes = 0;
        HIP_TRY(hipMalloc(&g->d_scratch, need));
        g->scratch_bytes = need;
      }
    }
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  if (use_wide && plan && !plan->verts.empty()) {
    // wide vertices first (the heaviest work of the launch): per round, phase 1 = the mining kernel over the row-group
    // chunks, phase 2 = the big-LDS count kernels per class; then the mining kernel over everything else
    my_edges += plan->edges;
    HIP_TRY(hipMemsetAsync(g->d_wide_queue, 0, 65536, stream));
    const bool prof = getenv("GM_WIDE_PROFILE") != nullptr;
    unsigned long long *d_prof = nullptr;
    if (prof) {
      HIP_TRY(hipMalloc(&d_prof, 4 * 32));
      HIP_TRY(hipMemset(d_prof, 0, 4 * 32));
    }
    int qword = 0;
    for (const auto &rd : plan->rounds) {
      if (qword + 4 > 16384) return GM_ERR_TOO_LARGE;  // (more than 4096 arena rounds)
      CliqueBuildParams pw;
      memset(&pw, 0, sizeof pw);
      pw.g = p.g;
      pw.chunks = plan->d_chunks + rd.chunk_begin;
      pw.count = (int)(rd.chunk_end - rd.chunk_begin);
      pw.queue = g->d_wide_queue + qword++;
      pw.mat = g->d_wide_mat;
      pw.base = plan->d_base;
      pw.cost_x_step = p.cost_x_step; pw.cost_y_step = p.cost_y_step; pw.cost_x_base = p.cost_x_base; pw.cost_y_base = p.cost_y_base;
      pw.flags = p.flags;
      const int per_cu_b = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / clique_build_lds_bytes()));
      const int wgrid = (int)std::max<long long>(1, std::min<long long>(pw.count, (long long)g->cu_count * per_cu_b));
      if (pw.count > 0) HIP_TRY(launch_clique_build(pw, wgrid, stream));
      for (int cls = 2; cls >= 0; --cls) {  // X and L (one workgroup per CU) before S
        CliqueCountParams c;
        memset(&c, 0, sizeof c);
        c.rp = g->d_rp;
        c.verts = plan->d_verts;
        c.base = plan->d_base;
        c.mat = g->d_wide_mat;
        c.slots = plan->d_cls_slots + rd.cls_begin[cls];
        c.count = (int)(rd.cls_begin[cls + 1] - rd.cls_begin[cls]);
        c.queue = g->d_wide_queue + qword++;
        c.counters = g->d_counters;
        c.profile = prof ? d_prof + 4 * cls : nullptr;
        if (c.count == 0) continue;
        const int per_cu_c = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / clique_count_lds_bytes(cls), (size_t)(2048 / clique_count_threads(cls))));
        const int cgrid = (int)std::max<long long>(1, std::min<long long>(c.count, (long long)g->cu_count * per_cu_c));
        HIP_TRY(launch_clique_count(cls, c, cgrid, stream));
      }
    }
    if (prof) {
      unsigned long long h[12];
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost));
      (void)hipFree(d_prof);
      for (int cls = 0; cls < 3; ++cls)
        if (h[4 * cls + 3])
          fprintf(stderr, "[wide] count class %c: %llu workgroups; per workgroup ms: load %.2f count %.2f\n", "SLX"[cls], h[4 * cls + 3],
                  h[4 * cls] / (double)h[4 * cls + 3] / 1e5, h[4 * cls + 1] / (double)h[4 * cls + 3] / 1e5);
      fprintf(stderr, "[wide] %zu vertices, %zu row-group chunks, %zu round(s), arena %.1f MB\n", plan->verts.size(), plan->n_chunks,
              plan->rounds.size(), g->wide_mat_bytes / 1048576.0);
    }
  }
  uint64_t chunks_total = (uint64_t)p.count;
  bool joined[3] = {false, false, false};
  if (use_classes) {
    // A class kernel whose share of chunks cannot fill the chip on its own (small graphs, 1/8 shares) runs on a side stream, so
    // that the kernels launched after it fill the idle CUs; one that can fill it stays on the launch's stream -- there every
    // kernel has the chip to itself (side streams at R-MAT-24 size: diamond 667 vs 636 ms, the 148 KB workgroups of class 2 wait
    // for whole CUs to drain). GM_CLASSES_STREAMS=0 / 1 forces one or the other.
    const char *streams_env = getenv("GM_CLASSES_STREAMS");
    auto side_stream = [&](int cls, long long count, long long full_grid, hipStream_t *ws) -> int {
      *ws = stream;
      const bool side = streams_env ? atoi(streams_env) != 0 : count < full_grid;
      if (!side) return GM_OK;
      *ws = g->aux_stream[cls - 1];
      HIP_TRY(hipStreamWaitEvent(*ws, ctx.evp[0], 0));  // after the counters were zeroed and the timer started
      return GM_OK;
    };
    auto side_done = [&](int cls, hipStream_t ws) -> int {
      if (ws == stream) return GM_OK;
      HIP_TRY(hipEventRecord(g->aux_done[cls - 1], ws));
      joined[cls - 1] = true;
      return GM_OK;
    };
    for (int cls = 3; cls >= 1; --cls) {
      if (!tab_cls[cls]) continue;
      MineParams q = p;
      take_share(tab_cls[cls], q);
      q.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + cls;  // its own dequeue word inside the zeroed 64-byte block
      if (q.count == 0) continue;
      chunks_total += (uint64_t)q.count;
      // the row as a hashed set in LDS (gm_hrow.hip) unless the ids are too wide for its 14-bit remainders
      // (tune[6] & 0x400000: A/B switch, the sorted LDS copy + bit filter + bisection of gm_mine_wide.hip)
      if (cls == 3) {
        const int rgrid = (int)std::max<long long>(1, std::min<long long>(q.count, (long long)g->cu_count * giant_per_cu()));
        const unsigned long long slot_words = giant_scratch_words(g->max_deg);  // (allocated before the timer started)
        q.scratch = g->d_scratch;
        q.scratch_words = slot_words;
        hipStream_t ws;
        rc = side_stream(cls, q.count, (long long)g->cu_count * giant_per_cu(), &ws);
        if (rc) return rc;
        HIP_TRY(launch_giant(pat, q, rgrid, ws));
        rc = side_done(cls, ws);
        if (rc) return rc;
        continue;
      }
      const bool hrow = !(la->tune[6] & 0x400000) && p.g.edesc != nullptr && hrow_fits(g->nv, cls);
      const int per_cu_w = hrow ? hrow_per_cu(cls) : (int)std::max<size_t>(1, (160 * 1024) / mine_wide_lds_bytes(cls));
      const int wgrid = (int)std::max<long long>(1, std::min<long long>(q.count, (long long)g->cu_count * per_cu_w));
      hipStream_t ws;
      rc = side_stream(cls, q.count, (long long)g->cu_count * per_cu_w, &ws);
      if (rc) return rc;
      if (hrow) HIP_TRY(launch_hrow(pat, cls, q, wgrid, ws));
      else HIP_TRY(launch_mine_wide(pat, cls, q, wgrid, ws));
      rc = side_done(cls, ws);
      if (rc) return rc;
    }
  }
  if (tab_long) {  // TC: the out-edges of the rows beyond the stage, through the chunked kernel (own dequeue word)
    MineParams q = p;
    take_share(tab_long, q);
    q.g.trp = nullptr;
    q.g.tdesc = nullptr;
    q.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 1;
    if (q.count > 0) {
      chunks_total += (uint64_t)q.count;
      const long long wq = ((long long)q.count + (long long)q.grab - 1) / (long long)q.grab;
      HIP_TRY(launch_mine(pat, q, (int)std::max<long long>(1, std::min<long long>(wq, (long long)g->cu_count * per_cu)), stream));
    }
  }
  if (p.count > 0 && use_tct) HIP_TRY(launch_tct(p, tct_stage, (int)std::max<long long>(1, std::min<long long>(want, (long long)g->cu_count * tct_per_cu(tct_stage))), stream));
  else if (p.count > 0) HIP_TRY(launch_mine(pat, p, grid, stream));
#ifdef GM_DEBUG_CHUNKS
  {
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<unsigned long long> ticks(tab->n);
    HIP_TRY(hipMemcpy(ticks.data(), d_ticks, sizeof(unsigned long long) * tab->n, hipMemcpyDeviceToHost));
    (void)hipFree(d_ticks);
    std::vector<ChunkRec> recs(tab->n);
    HIP_TRY(hipMemcpy(recs.data(), tab->d, sizeof(ChunkRec) * tab->n, hipMemcpyDeviceToHost));
    std::vector<size_t> idx(tab->n);
    for (size_t i = 0; i < tab->n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ticks[a] > ticks[b]; });
    unsigned long long tot = 0;
    for (auto t : ticks) tot += t;
    if (const char *dump = getenv("GM_CHUNK_DUMP")) {
      static int dump_no = 0;
      const std::string name = std::string(dump) + "." + std::to_string(dump_no++) + ".pat" + std::to_string((int)pat);
      FILE *f = fopen(name.c_str(), "w");
      if (f) {
        fprintf(f, "pos,cid,rows,entries,rowlen,whole,part,nparts,cost,us\n");
        for (size_t pos = 0; pos < tab->n; ++pos) {
          const size_t cid = p.order ? (size_t)(p.order == tab->d_order[0] ? tab->order[0][pos] : tab->order[1][pos]) : pos;
          const ChunkRec &r = recs[cid];
          const bool whole = r.e_begin == g->h_rp[r.u_begin] && r.e_end == g->h_rp[r.u_end];
          fprintf(f, "%zu,%zu,%d,%d,%d,%d,%d,%d,%llu,%.1f\n", pos, cid, r.u_end - r.u_begin, r.e_end - r.e_begin,
                  g->h_rp[r.u_begin + 1] - g->h_rp[r.u_begin], (int)whole, r.part, r.nparts, tab->cost[cid], ticks[pos] / 100.0);
        }
        fclose(f);
      }
    }
    fprintf(stderr, "[chunks] n=%zu total ticks %llu (100 MHz): mean %.1f us\n", tab->n, tot, tot / 100.0 / std::max<size_t>(tab->n, 1));
    for (size_t k = 0; k < std::min<size_t>(12, tab->n); ++k) {
      const size_t pos = idx[k];
      const size_t cid = p.order ? (size_t)(p.order == tab->d_order[0] ? tab->order[0][pos] : tab->order[1][pos]) : pos;
      const ChunkRec &r = recs[cid];
      fprintf(stderr, "[chunks] #%zu pos %zu: %.1f us  rows [%d,%d) entries [%d,%d) n=%d part %d/%d rowlen %d\n", k, pos, ticks[pos] / 100.0,
              r.u_begin, r.u_end, r.e_begin, r.e_end, r.e_end - r.e_begin, r.part, r.nparts, g->h_rp[r.u_begin + 1] - g->h_rp[r.u_begin]);
    }
  }
#endif
  for (int i = 0; i < 3; ++i)
    if (joined[i]) HIP_TRY(hipStreamWaitEvent(stream, g->aux_done[i], 0));  // the launch ends when all three kernels have
  fill_stats(st, (pat == PAT_DIAMOND || pat == PAT_MOTIF4E) ? my_edges / 2 : my_edges, chunks_total, grid,
             kWavesPerBlock * GM_WAVE);
  return end_launch(ctx, fin_mode, fin_base, h_out, nout, st);
}

extern "C" int gm_kernel_times(const gm_graph *g, int n, double *ms_out, int *n_out) {
  if (!g || !ms_out || !n_out || n < 0) return GM_ERR_INVALID;
  if (g->ring_alias) g = g->ring_alias;
  const unsigned long long have = std::min<unsigned long long>(g->ev_launches, gm_graph::kEvRing);
  const int m = (int)std::min<unsigned long long>((unsigned long long)n, have);
  HIP_TRY(hipSetDevice(g->device));
  for (int i = 0; i < m; ++i) {  // oldest of the last m launches first
    const unsigned long long idx = (g->ev_launches - (unsigned long long)m + (unsigned long long)i) % gm_graph::kEvRing;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[idx][0], g->ev[idx][1]));
    ms_out[i] = ms;
  }
  *n_out = m;
  return GM_OK;
}

extern "C" int gm_tc(const gm_graph *dag, const gm_launch *la, uint64_t *total, gm_stats *st) {
  return run_pattern(PAT_TC, dag, la, 3, total, 1, st);
}

// rectangle, flattened over wedges (rect_flat_kernel in gm_mine.hip)
static int ensure_idx0(gm_graph *g, const GraphView &gv) {
  if (g->d_idx0) return GM_OK;
  OtherSetupScope scope(g);
  HIP_TRY(hipMalloc(&g->d_idx0, sizeof(int) * (size_t)std::max(g->nv, 1)));
  HIP_TRY(launch_idx0(gv, g->d_idx0, 0));
  return GM_OK;
}

static int run_rect_flat(const gm_graph *cg, const gm_launch *la_in, uint64_t *h_out, gm_stats *st, bool pentagon = false) {
  LaunchCtx ctx;
  int rc = begin_launch(cg, la_in, h_out, ctx);
  if (rc) return rc;
  gm_graph *g = ctx.g;
  const gm_launch *la = &ctx.la;
  GraphView gv;
  gv.nv = g->nv;
  gv.ne = (int)g->ne;
  gv.rp = g->d_rp;
  gv.col = g->d_col;
  if (!g->d_wblock_prefix) {  // once per graph: idx0[v] on the device, wedge-block prefix on the host
    OtherSetupScope scope(g);
    rc = ensure_idx0(g, gv);
    if (rc) return rc;
    std::vector<int> idx0((size_t)std::max(g->nv, 1));
    HIP_TRY(hipMemcpy(idx0.data(), g->d_idx0, sizeof(int) * (size_t)g->nv, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> pre((size_t)g->nv + 1);
    unsigned long long acc = 0;
    for (int v = 0; v < g->nv; ++v) {
      pre[v] = acc;
      const unsigned long long n = (unsigned long long)idx0[v];
      acc += (n * (n - (n ? 1ull : 0ull)) / 2ull + 63ull) / 64ull;
    }
    pre[g->nv] = acc;
    g->n_wblocks = acc;
    HIP_TRY(hipMalloc(&g->d_wblock_prefix, sizeof(unsigned long long) * ((size_t)g->nv + 1)));
    HIP_TRY(hipMemcpy(g->d_wblock_prefix, pre.data(), sizeof(unsigned long long) * ((size_t)g->nv + 1), hipMemcpyHostToDevice));
  }
  RectParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.idx0 = g->d_idx0;
  p.block_prefix = g->d_wblock_prefix;
  p.nblocks = g->n_wblocks;
  p.group = la->chunk > 0 ? la->chunk : (pentagon ? 2 : 16);
  const long long ngroups = (long long)((p.nblocks + (unsigned long long)p.group - 1) / (unsigned long long)p.group);
  int64_t first = 0, step = 1, count = 0;
  gm_partition(ngroups, ctx.rank, ctx.world, la->policy, &first, &step, &count);
  p.first = (unsigned long long)first;
  p.step = (unsigned long long)step;
  p.count = (unsigned long long)count;
  p.counters = g->d_counters;
  p.queue = g->d_counters + 4;
  const int grid = (int)std::max<long long>(1, std::min<long long>((count + 3) / 4, (long long)g->cu_count * 8));
  rc = start_timer(ctx);
  if (rc) return rc;
  if (count > 0) HIP_TRY(launch_rect_flat(p, pentagon, grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

static int ensure_edge_tables(gm_graph *g, const GraphView &gv);

// rectangle (rect_acc_kernel) and pentagon (pent_acc_kernel) by wedge accumulation: same centres, same counter maps
static int run_rect_acc(const gm_graph *cg, const gm_launch *la_in, uint64_t *h_out, gm_stats *st, bool pentagon = false) {
  LaunchCtx ctx;
  int rc = begin_launch(cg, la_in, h_out, ctx);
  if (rc) return rc;
  gm_graph *g = ctx.g;
  const gm_launch *la = &ctx.la;
  GraphView gv;
  gv.nv = g->nv;
  gv.ne = (int)g->ne;
  gv.rp = g->d_rp;
  gv.col = g->d_col;
  rc = ensure_idx0(g, gv);
  if (rc) return rc;
  if (!g->d_rect_tasks) {  // once per graph: 2-path estimate per centre (device), task list (host): heavy first, then light by 4
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long *d_work = nullptr;
    HIP_TRY(hipMalloc(&d_work, sizeof(unsigned long long) * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1));
    hipError_t e = nv ? launch_rect_work(gv, g->d_idx0, d_work, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
    (void)hipFree(d_work);
    if (e != hipSuccess) return hip_fail(e, "rect_work_kernel", __FILE__, __LINE__);
    std::vector<int> vs;
    vs.reserve(nv);
    for (size_t v = 0; v < nv; ++v)
      if (work[v] > 0) vs.push_back((int)v);
    std::stable_sort(vs.begin(), vs.end(), [&](int a, int b) { return work[(size_t)a] > work[(size_t)b]; });
    const unsigned long long heavy = 1ull << 15;  // 2-paths above which a centre gets a whole workgroup
    std::vector<int4> tasks;
    size_t i = 0;
    for (; i < vs.size() && work[(size_t)vs[i]] >= heavy; ++i) tasks.push_back(make_int4(vs[i], -2, -2, -2));
    for (; i < vs.size(); i += 4) {
      int4 t = make_int4(-1, -1, -1, -1);
      t.x = vs[i];
      if (i + 1 < vs.size()) t.y = vs[i + 1];
      if (i + 2 < vs.size()) t.z = vs[i + 2];
      if (i + 3 < vs.size()) t.w = vs[i + 3];
      tasks.push_back(t);
    }
    g->n_rect_tasks = tasks.size();
    HIP_TRY(hipMalloc(&g->d_rect_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_rect_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
  }
  RectAccParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.idx0 = g->d_idx0;
  p.tasks = g->d_rect_tasks;
  int64_t first = 0, step = 1, count = 0;
  gm_partition((int64_t)g->n_rect_tasks, ctx.rank, ctx.world, la->policy, &first, &step, &count);
  p.first = (unsigned long long)first;
  p.step = (unsigned long long)step;
  p.count = (unsigned long long)count;
  p.counters = g->d_counters;
  p.queue = g->d_counters + 4;
  // one counter map (nv words) per wave, within a memory budget
  p.acc_stride = ((unsigned long long)g->nv + 63ull) & ~63ull;
  const unsigned long long per_wg = p.acc_stride * 4ull * kWavesPerBlock;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const unsigned long long budget = std::min<unsigned long long>(32ull << 30, (unsigned long long)free_b / 4 + (unsigned long long)g->rect_acc_bytes);  // (maps + touched lists)
  long long grid = std::min<long long>((long long)g->cu_count * 8, (long long)std::max<unsigned long long>(1, budget / std::max<unsigned long long>(per_wg, 1)));
  grid = std::max<long long>(1, std::min<long long>(grid, count));
  const size_t need = (size_t)per_wg * (size_t)grid;
  if (need > g->rect_acc_bytes) {
    if (g->d_rect_acc) (void)hipFree(g->d_rect_acc);
    g->d_rect_acc = nullptr;
    g->rect_acc_bytes = 0;
    HIP_TRY(hipMalloc(&g->d_rect_acc, need));
    HIP_TRY(hipMemset(g->d_rect_acc, 0, need));  // every launch leaves the maps zeroed again
    g->rect_acc_bytes = need;
  }
  p.acc = g->d_rect_acc;
  if (need > g->pent_touched_bytes) {  // touched-vertex lists: same shape as the maps (one int list of up to nv entries per wave)
    if (g->d_pent_touched) (void)hipFree(g->d_pent_touched);
    g->d_pent_touched = nullptr;
    g->pent_touched_bytes = 0;
    HIP_TRY(hipMalloc(&g->d_pent_touched, need));
    g->pent_touched_bytes = need;
  }
  p.touched = g->d_pent_touched;
  if (pentagon) {
    rc = ensure_edge_tables(g, gv);
    if (rc) return rc;
    PentAccParams q;
    memset(&q, 0, sizeof q);
    q.g = gv;
    q.idx0 = g->d_idx0;
    q.tlt = g->d_house_tlt;
    q.tasks = p.tasks;
    q.first = p.first;
    q.step = p.step;
    q.count = p.count;
    q.acc = p.acc;
    q.touched = g->d_pent_touched;
    q.acc_stride = p.acc_stride;
    q.queue = p.queue;
    q.counters = p.counters;
    rc = start_timer(ctx);
    if (rc) return rc;
    if (count > 0) HIP_TRY(launch_pent_acc(q, (int)grid, ctx.stream));
    fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, (int)grid, 256);
    return end_launch(ctx, FIN_HALF_SIGNED, 0, h_out, 1, st);
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  if (count > 0) HIP_TRY(launch_rect_acc(p, (int)grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, (int)grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

// per-entry triangle tables t / tlt (edge_tab_kernel), once per graph; every rank builds the whole tables: they are inputs of
// every centre of the house / pentagon map kernels
static int ensure_edge_tables(gm_graph *g, const GraphView &gv) {
  if (g->d_house_t && g->d_house_tlt) return GM_OK;
  OtherSetupScope scope(g);
  const size_t ne1 = (size_t)std::max<long long>(g->ne, 1);
  DevBuf<unsigned> t, tlt;
  HIP_TRY(t.alloc(ne1));
  HIP_TRY(tlt.alloc(ne1));
  if (g->ne > 0) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(g->d_counters, 0, 64));
    HIP_TRY(launch_edge_tab(gv, t.p, tlt.p, g->d_counters + 4, g->cu_count * 8, 0));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(g->d_counters, 0, 64));  // (the table kernel used the dequeue head)
  }
  g->d_house_t = t.release();  // (published only after the build kernel has succeeded; the buffers free themselves on the error paths)
  g->d_house_tlt = tlt.release();
  return GM_OK;
}

// house by wedge accumulation (edge_tab_kernel + house_acc_kernel in gm_mine.hip)
static int run_house_acc(const gm_graph *cg, const gm_launch *la_in, uint64_t *h_out, gm_stats *st) {
  LaunchCtx ctx;
  int rc = begin_launch(cg, la_in, h_out, ctx);
  if (rc) return rc;
  gm_graph *g = ctx.g;
  const gm_launch *la = &ctx.la;
  GraphView gv;
  gv.nv = g->nv;
  gv.ne = (int)g->ne;
  gv.rp = g->d_rp;
  gv.col = g->d_col;
  rc = ensure_edge_tables(g, gv);
  if (rc) return rc;
  if (!g->d_house_tasks) {  // once per graph
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long *d_work = nullptr;
    HIP_TRY(hipMalloc(&d_work, sizeof(unsigned long long) * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1));
    hipError_t e = nv ? launch_house_work(gv, d_work, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
    (void)hipFree(d_work);
    if (e != hipSuccess) return hip_fail(e, "house_work_kernel", __FILE__, __LINE__);
    std::vector<int> vs;
    vs.reserve(nv);
    for (size_t v = 0; v < nv; ++v)
      if (work[v] > 0) vs.push_back((int)v);
    std::stable_sort(vs.begin(), vs.end(), [&](int a, int b) { return work[(size_t)a] > work[(size_t)b]; });
    const unsigned long long heavy = 1ull << 15;
    std::vector<int4> tasks;
    size_t i = 0;
    for (; i < vs.size() && work[(size_t)vs[i]] >= heavy; ++i) tasks.push_back(make_int4(vs[i], -2, -2, -2));
    for (; i < vs.size(); i += 4) {
      int4 t4 = make_int4(-1, -1, -1, -1);
      t4.x = vs[i];
      if (i + 1 < vs.size()) t4.y = vs[i + 1];
      if (i + 2 < vs.size()) t4.z = vs[i + 2];
      if (i + 3 < vs.size()) t4.w = vs[i + 3];
      tasks.push_back(t4);
    }
    g->n_house_tasks = tasks.size();
    HIP_TRY(hipMalloc(&g->d_house_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_house_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
  }
  HouseAccParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.t = g->d_house_t;
  p.tlt = g->d_house_tlt;
  p.tasks = g->d_house_tasks;
  int64_t first = 0, step = 1, count = 0;
  gm_partition((int64_t)g->n_house_tasks, ctx.rank, ctx.world, la->policy, &first, &step, &count);
  p.first = (unsigned long long)first;
  p.step = (unsigned long long)step;
  p.count = (unsigned long long)count;
  p.counters = g->d_counters;
  p.queue = g->d_counters + 4;
  p.acc_stride = ((unsigned long long)g->nv + 63ull) & ~63ull;
  const unsigned long long per_wg = p.acc_stride * 8ull * kWavesPerBlock;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const unsigned long long budget = std::min<unsigned long long>(32ull << 30, (unsigned long long)free_b / 3 + (unsigned long long)g->house_acc_bytes);  // (maps + touched lists)
  long long grid = std::min<long long>((long long)g->cu_count * 8, (long long)std::max<unsigned long long>(1, budget / std::max<unsigned long long>(per_wg, 1)));
  grid = std::max<long long>(1, std::min<long long>(grid, count));
  const size_t need = (size_t)per_wg * (size_t)grid;
  if (need > g->house_acc_bytes) {
    if (g->d_house_acc) (void)hipFree(g->d_house_acc);
    g->d_house_acc = nullptr;
    g->house_acc_bytes = 0;
    HIP_TRY(hipMalloc(&g->d_house_acc, need));
    HIP_TRY(hipMemset(g->d_house_acc, 0, need));  // every launch leaves the maps zeroed again
    if (g->d_house_touched) (void)hipFree(g->d_house_touched);
    g->d_house_touched = nullptr;
    HIP_TRY(hipMalloc(&g->d_house_touched, need / 2));  // one int list per map
    g->house_acc_bytes = need;
  }
  p.acc = g->d_house_acc;
  p.touched = g->d_house_touched;
  rc = start_timer(ctx);
  if (rc) return rc;
  if (count > 0) HIP_TRY(launch_house_acc(p, (int)grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, (int)grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

// house, flattened over (v0, v1, v3) tasks (house_flat_kernel in gm_mine.hip)
static int run_house_flat(const gm_graph *cg, const gm_launch *la_in, uint64_t *h_out, gm_stats *st) {
  LaunchCtx ctx;
  int rc = begin_launch(cg, la_in, h_out, ctx);
  if (rc) return rc;
  gm_graph *g = ctx.g;
  const gm_launch *la = &ctx.la;
  GraphView gv;
  gv.nv = g->nv;
  gv.ne = (int)g->ne;
  gv.rp = g->d_rp;
  gv.col = g->d_col;
  if (!g->d_house_prefix) {  // once per graph: blocks per entry on the device, prefix on the host
    OtherSetupScope scope(g);
    const size_t ne = (size_t)g->ne;
    unsigned *d_nblk = nullptr;
    HIP_TRY(hipMalloc(&d_nblk, sizeof(unsigned) * std::max<size_t>(ne, 1)));
    std::vector<unsigned> nblk(std::max<size_t>(ne, 1));
    hipError_t e = ne ? launch_house_blocks(gv, d_nblk, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(nblk.data(), d_nblk, sizeof(unsigned) * ne, hipMemcpyDeviceToHost);
    (void)hipFree(d_nblk);
    if (e != hipSuccess) return hip_fail(e, "house block table", __FILE__, __LINE__);
    std::vector<unsigned long long> pre(ne + 1);
    unsigned long long acc = 0;
    for (size_t i = 0; i < ne; ++i) { pre[i] = acc; acc += nblk[i]; }
    pre[ne] = acc;
    g->n_house_blocks = acc;
    HIP_TRY(hipMalloc(&g->d_house_prefix, sizeof(unsigned long long) * (ne + 1)));
    HIP_TRY(hipMemcpy(g->d_house_prefix, pre.data(), sizeof(unsigned long long) * (ne + 1), hipMemcpyHostToDevice));
  }
  HouseParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.entry_prefix = g->d_house_prefix;
  p.nblocks = g->n_house_blocks;
  p.group = la->chunk > 0 ? la->chunk : 8;
  p.no_bits = (la->tune[6] & 0x8000) ? 1 : 0;
  const long long ngroups = (long long)((p.nblocks + (unsigned long long)p.group - 1) / (unsigned long long)p.group);
  int64_t first = 0, step = 1, count = 0;
  gm_partition(ngroups, ctx.rank, ctx.world, la->policy, &first, &step, &count);
  p.first = (unsigned long long)first;
  p.step = (unsigned long long)step;
  p.count = (unsigned long long)count;
  p.counters = g->d_counters;
  p.queue = g->d_counters + 4;
  const int grid = (int)std::max<long long>(1, std::min<long long>((count + 3) / 4, (long long)g->cu_count * 8));
  rc = start_timer(ctx);
  if (rc) return rc;
  if (count > 0 && g->ne > 0) HIP_TRY(launch_house_flat(p, grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

// rectangle / house / pentagon: one wave per symmetry-broken edge (gm_sgl.hip)
static int run_sgl_nested(int pat, const gm_graph *cg, const gm_launch *la_in, uint64_t *h_out, gm_stats *st) {
  LaunchCtx ctx;
  int rc = begin_launch(cg, la_in, h_out, ctx);
  if (rc) return rc;
  gm_graph *g = ctx.g;
  const gm_launch *la = &ctx.la;
  SglParams p;
  memset(&p, 0, sizeof p);
  p.g.nv = g->nv;
  p.g.ne = (int)g->ne;
  p.g.rp = g->d_rp;
  p.g.col = g->d_col;
  p.chunk = la->chunk > 0 ? la->chunk : 64;
  const long long nchunks = (g->ne + p.chunk - 1) / p.chunk;
  int64_t first = 0, step = 1, count = 0;
  gm_partition(nchunks, ctx.rank, ctx.world, la->policy, &first, &step, &count);
  p.first = first;
  p.step = step;
  p.count = count;
  p.counters = g->d_counters;
  p.queue = reinterpret_cast<unsigned *>(g->d_counters + 4);
  p.max_deg = std::max(g->max_deg, 1);
  const int grid = (int)std::max<long long>(1, std::min<long long>((count + 3) / 4, (long long)g->cu_count * 8));
  if (pat == SGL_HOUSE || pat == SGL_DIAMOND) {  // per-wave list for the materialised S = N(v0) ^ N(v1)
    const size_t need = (size_t)grid * 4 * (size_t)p.max_deg * sizeof(int);
    if (need > g->scratch_bytes) {
      if (g->d_scratch) (void)hipFree(g->d_scratch);
      g->d_scratch = nullptr;
      g->scratch_bytes = 0;
      HIP_TRY(hipMalloc(&g->d_scratch, need));
      g->scratch_bytes = need;
    }
    p.scratch = reinterpret_cast<int *>(g->d_scratch);
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  if (count > 0) HIP_TRY(launch_sgl_nested(pat, p, grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)count, grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

extern "C" int gm_sgl(const gm_graph *sym, const char *pattern, const gm_launch *la, uint64_t *total, gm_stats *st) {
  if (!pattern) return GM_ERR_INVALID;
  if (strcmp(pattern, "diamond") == 0) {
    // tune[6] & 1024: the LISTING (nested) form of the reference, src/sgl/gpu_kernels/diamond_nested.cuh:4-31 -- materialise
    // S, count_smaller per member -- as a second implementation; the default counts C(|S|,2) per edge (diamond_count.cuh:15-17)
    if (la && (la->tune[6] & 1024)) return run_sgl_nested(SGL_DIAMOND, sym, la, total, st);
    return run_pattern(PAT_DIAMOND, sym, la, 4, total, 1, st);
  }
  // rectangle / house / pentagon run on a copy of the graph renumbered by degree (get_relabeled; tune[6] & 512: on the
  // graph as given). tune[6] & 1024: the wave-per-edge loop nests; & 2048: rectangle as wedges + flat intersections,
  // house without the LDS S-bitmap (A/B, tests).
  const bool is_rect = strcmp(pattern, "rectangle") == 0, is_house = strcmp(pattern, "house") == 0, is_pent = strcmp(pattern, "pentagon") == 0;
  if (is_rect || is_house || is_pent) {
    if (!sym) return GM_ERR_INVALID;
    const int t6 = la ? la->tune[6] : 0;
    gm_graph *self = const_cast<gm_graph *>(sym);
    const gm_graph *run_on = sym;
    const bool wedge_form = (is_pent || is_rect) && (t6 & 2048);  // anchored wedges: hubs first; 2-path / (v0,v1,v3) forms: hubs last
    // house: by wedge accumulation (default; its 2-path count does not depend on the numbering, so no renumbered copy), or
    // the flattened (v0, v1, v3) form (0x800; 0x8000: without the LDS S-bitmap). The packed map holds 24-bit counts and
    // 40-bit weighted sums: rows of 2^20 entries or more take the flattened form.
    const bool house_acc = is_house && !(t6 & (1024 | 2048 | 0x8000)) && sym->max_deg < (1 << 20);
    if (!(t6 & 512) && !(t6 & 1024) && !house_acc) {
      gm_graph *r = nullptr;
      int rc = get_relabeled(self, wedge_form ? 1 : 0, &r);
      if (rc) return rc;
      run_on = r;
    }
    int rc;
    if (t6 & 1024) rc = run_sgl_nested(is_rect ? SGL_RECTANGLE : is_house ? SGL_HOUSE : SGL_PENTAGON, run_on, la, total, st);
    else if (house_acc) rc = run_house_acc(run_on, la, total, st);
    else if (is_house) rc = run_house_flat(run_on, la, total, st);
    else if (is_pent && (t6 & 2048)) rc = run_rect_flat(run_on, la, total, st, true);
    else if (is_pent) rc = run_rect_acc(run_on, la, total, st, true);
    else if (t6 & 2048) rc = run_rect_flat(run_on, la, total, st);
    else rc = run_rect_acc(run_on, la, total, st);
    self->ring_alias = (run_on != sym) ? const_cast<gm_graph *>(run_on) : nullptr;
    return rc;
  }
  if (total) *total = 0;  // "Not implemented", total_num = 0 (src/sgl/omp_base.cc:51-53)
  return GM_ERR_UNSUPPORTED;
}

extern "C" int gm_clique(const gm_graph *dag, int k, const gm_launch *la, uint64_t *total, gm_stats *st) {
  if (k == 3) return run_pattern(PAT_TC, dag, la, 3, total, 1, st);
  if (k < 3 || k > 8) {
    if (total) *total = 0;
    return GM_ERR_INVALID;
  }
  if (k > 4 && dag && dag->max_deg > 4096) {  // deeper levels sweep a row with two words per lane (cliquek_count_sub)
    if (total) *total = 0;
    g_last_error = "gm_clique: k >= 5 needs max out-degree <= 4096 (this DAG: " + std::to_string(dag->max_deg) + ")";
    return GM_ERR_TOO_LARGE;
  }
  return run_pattern(k == 4 ? PAT_CLIQUE4 : PAT_CLIQUEK, dag, la, k, total, 1, st);
}

// 4-motif, formula form (src/motif/cpu_kernels/automine_formula.h:21-56 + src/motif/omp_formula.cc:41-45):
// raw[0..3] = the per-edge sums counter[0], counter[1], counter[2], counter[4] (PAT_MOTIF4E, one |N(v0)^N(v1)| per
// undirected edge), raw[4] = edge-induced 4-cycles (rectangle kernel), raw[5] = 4-cliques (clique kernel on the cached
// DAG). Every raw value is a plain sum over tasks, so per-rank partials add up; gm_motif4_finish turns the summed raw
// values into the six vertex-induced counts.
// With la->d_counts set the six raw sums are left in that DEVICE buffer (raw[0..3] by the per-edge kernel, raw[4] by the
// rectangle kernel, raw[5] by the clique kernel, all ordered on la->stream) and nothing synchronises unless `raw` is given
// too -- this is what feeds the RCCL all-reduce of motif_multigpu.
extern "C" int gm_motif4_partial(const gm_graph *sym, const gm_launch *la, uint64_t raw[6], gm_stats *st) {
  if (!sym || (!raw && !(la && la->d_counts))) return GM_ERR_INVALID;
  gm_graph *g = const_cast<gm_graph *>(sym);
  gm_launch l2;
  memset(&l2, 0, sizeof l2);
  if (la) l2 = *la;
  uint64_t *d_out = l2.d_counts;
  if (!g->dag_cache) {
    gm_graph *dag = nullptr;
    int rc = gm_graph_orient(sym, &dag);
    if (rc) return rc;
    g->dag_cache = dag;
  }
  gm_stats s1, s2, s3;
  memset(&s1, 0, sizeof s1); memset(&s2, 0, sizeof s2); memset(&s3, 0, sizeof s3);
  l2.d_counts = d_out;
  int rc = run_pattern(PAT_MOTIF4E, sym, &l2, 4, raw, 4, &s1, FIN_RAW4, 0);
  if (rc) return rc;
  {
    const gm_graph *rect_on = sym;
    if (!(l2.tune[6] & 512)) {
      gm_graph *r = nullptr;
      rc = get_relabeled(g, (l2.tune[6] & 2048) ? 1 : 0, &r);
      if (rc) return rc;
      rect_on = r;
    }
    l2.d_counts = d_out ? d_out + 4 : nullptr;
    rc = (l2.tune[6] & 2048) ? run_rect_flat(rect_on, &l2, raw ? &raw[4] : nullptr, &s2)
                             : run_rect_acc(rect_on, &l2, raw ? &raw[4] : nullptr, &s2);
  }
  if (rc) return rc;
  l2.d_counts = d_out ? d_out + 5 : nullptr;
  rc = run_pattern(PAT_CLIQUE4, g->dag_cache, &l2, 4, raw ? &raw[5] : nullptr, 1, &s3);
  if (rc) return rc;
  if (st) {
    *st = s1;
    st->kernel_ms = s1.kernel_ms + s2.kernel_ms + s3.kernel_ms;
  }
  return GM_OK;
}

// raw sums -> the six vertex-induced counts (host fix-up of src/motif/omp_formula.cc:41-45, same arithmetic on both sides)
__host__ __device__ static inline void motif4_finish_math(const unsigned long long *raw, unsigned long long *counts) {
  const unsigned long long k4 = raw[5];
  const unsigned long long diamond = raw[3] / 2 - 6 * k4;            // total[4] = total[4]/2 - 6*total[5]
  const unsigned long long tailed = raw[2] / 2 - 2 * diamond;        // total[2] = total[2]/2 - 2*total[4]
  const unsigned long long cycle4 = raw[4] - diamond - 3 * k4;       // vertex-induced 4-cycles from the edge-induced count
  const unsigned long long path4 = raw[1] - 4 * cycle4;              // total[1] = total[1] - 4*total[3]
  const unsigned long long star3 = raw[0] / 6 - tailed / 3;          // total[0] = total[0]/6 - total[2]/3
  counts[0] = star3; counts[1] = path4; counts[2] = tailed; counts[3] = cycle4; counts[4] = diamond; counts[5] = k4;
}

__global__ void motif4_finish_kernel(unsigned long long *__restrict__ c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long raw[6], out[6];
  for (int i = 0; i < 6; ++i) raw[i] = c[i];
  motif4_finish_math(raw, out);
  for (int i = 0; i < 6; ++i) c[i] = out[i];
}

extern "C" int gm_motif4_finish(const uint64_t raw[6], uint64_t counts[6]) {
  if (!raw || !counts) return GM_ERR_INVALID;
  unsigned long long r[6], c[6];
  for (int i = 0; i < 6; ++i) r[i] = raw[i];
  motif4_finish_math(r, c);
  for (int i = 0; i < 6; ++i) counts[i] = c[i];
  return GM_OK;
}

extern "C" int gm_motif(const gm_graph *sym, int k, const gm_launch *la, uint64_t *counts, int ncounts, gm_stats *st) {
  if (k == 4) {
    const bool async = la && la->d_counts;  // the asynchronous contract of every solver: counts may be NULL then
    if (ncounts < 6 || (!counts && !async)) return GM_ERR_INVALID;
    if (la && la->world > 1) return GM_ERR_UNSUPPORTED;  // multi-GPU: gm_motif4_partial + all-reduce + gm_motif4_finish
    uint64_t raw[6];
    int rc = gm_motif4_partial(sym, la, counts ? raw : nullptr, st);
    if (rc) return rc;
    if (async) {  // finish in place on the device, ordered on the launch stream
      hipLaunchKernelGGL(motif4_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)la->stream, (unsigned long long *)la->d_counts);
      HIP_TRY(hipGetLastError());
    }
    return counts ? gm_motif4_finish(raw, counts) : GM_OK;
  }
  if (k != 3) return GM_ERR_INVALID;
  if (ncounts < 2) return GM_ERR_INVALID;
  return run_pattern(PAT_MOTIF3, sym, la, 3, counts, ncounts, st);
}

// motif_omp_formula / motif_gpu_formula (src/motif/omp_formula.cc:39-46, cpu_kernels/automine_formula.h:2-19):
// enumerate only the triangles, derive the wedges: wedges = sum_v C(d(v),2) - 3*T. Here the triangles come from the
// TC kernel on the oriented graph (built once per handle and cached), so the hub rows of the symmetric graph are
// never intersected. Counts are identical to gm_motif; with world > 1 the sum_v C(d,2) term is contributed by rank 0
// and the per-rank partial wedge count is only meaningful after the all-reduce (mod 2^64 arithmetic).
__global__ __launch_bounds__(256) void sum_c2_kernel(int nv, const int *__restrict__ rp, unsigned long long *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
    const unsigned long long d = (unsigned long long)(rp[v + 1] - rp[v]);
    s += d * (d - 1) / 2;
  }
  s = gm::wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

extern "C" int gm_motif_formula(const gm_graph *sym, int k, const gm_launch *la, uint64_t *counts, int ncounts, gm_stats *st) {
  if (!sym) return GM_ERR_INVALID;
  if (k != 3) return (k == 4) ? GM_ERR_UNSUPPORTED : GM_ERR_INVALID;
  if (ncounts < 2) return GM_ERR_INVALID;
  gm_graph *g = const_cast<gm_graph *>(sym);
  if (!g->dag_cache) {
    gm_graph *dag = nullptr;
    int rc = gm_graph_orient(sym, &dag);
    if (rc) return rc;
    g->dag_cache = dag;
  }
  if (!g->sum_c2_valid) {  // sum_v C(d(v),2): one reduction kernel over the offsets
    DevBuf<unsigned long long> acc;
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(acc.alloc(1));
    HIP_TRY(hipMemset(acc.p, 0, 8));
    hipLaunchKernelGGL(sum_c2_kernel, dim3((unsigned)std::min<long long>(((long long)g->nv + 255) / 256, 2048)), dim3(256), 0, 0, g->nv, g->d_rp, acc.p);
    unsigned long long s2 = 0;
    HIP_TRY(hipMemcpy(&s2, acc.p, 8, hipMemcpyDeviceToHost));
    g->sum_c2 = s2;
    g->sum_c2_valid = true;
  }
  const int rank = la ? la->rank : 0;
  const int rc = run_pattern(PAT_TC, g->dag_cache, la, 3, counts, ncounts, st, FIN_MOTIF3_FORMULA, rank == 0 ? g->sum_c2 : 0ull);
  g->ring_alias = g->dag_cache;
  return rc;
}

// Tooling: the level-2 part of SURVEY.md 8(d)'s ALGORITHMIC bytes of one 4-clique launch,
//   sum_e [ 8|S1| + sum_{v2 in S1} (4(|S1| + d+(v2)) + 16) ]  =  24*sum|S1| + 4*sum|S1|^2 + 4*sum_{matches} d+(v2),
// from three sums the mining kernel itself produces (PAT_DAGSTATS). The level-1 part 4*sum_e(d+(v0)+d+(v1)) + 40|E+|
// is the TC formula (computed by the caller from the CSR arrays).
extern "C" int gm_clique4_level2_bytes(const gm_graph *dag, uint64_t *bytes) {
  if (!dag || !bytes) return GM_ERR_INVALID;
  uint64_t raw[4] = {0, 0, 0, 0};  // raw[0] = sum n^2, raw[1] = sum_{matches} d+(v2), raw[2] = sum n
  const int rc = run_pattern(PAT_DAGSTATS, dag, nullptr, 4, raw, 4, nullptr, FIN_RAW4, 0);
  if (rc) return rc;
  *bytes = 24ull * raw[2] + 4ull * raw[0] + 4ull * raw[1];
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// set-op batch (one wave per pair)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void setop_kernel(int op, long long npairs, const int *__restrict__ vals,
                                                    const long long *__restrict__ ab, const long long *__restrict__ ae,
                                                    const long long *__restrict__ bb, const long long *__restrict__ be,
                                                    const int *__restrict__ upper, const int *__restrict__ skip,
                                                    unsigned *__restrict__ out_num, int *__restrict__ out_vals) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long i = wave; i < npairs; i += nwaves) {
    const int *A = vals + ab[i];
    const int *B = vals + bb[i];
    const int a = (int)(ae[i] - ab[i]), b = (int)(be[i] - bb[i]);
    const int up = upper ? upper[i] : 0x7fffffff;
    const int sk = skip ? skip[i] : -1;
    int *O = out_vals ? out_vals + ab[i] : nullptr;
    unsigned r = 0;
    switch (op) {
      case GM_OP_INTERSECT_NUM: r = (unsigned)wave_sum((int)wave_intersect_num(A, a, B, b)); break;
      case GM_OP_INTERSECT_NUM_UPPER: r = (unsigned)wave_sum((int)wave_intersect_num_upper(A, a, B, b, up)); break;
      case GM_OP_INTERSECT_SET: r = (unsigned)wave_intersect_set(A, a, B, b, O); break;
      case GM_OP_INTERSECT_SET_UPPER: r = (unsigned)wave_intersect_set_upper(A, a, B, b, up, O); break;
      case GM_OP_DIFFERENCE_NUM: r = (unsigned)wave_sum((int)wave_difference_num(A, a, B, b, sk)); break;
      case GM_OP_DIFFERENCE_NUM_UPPER: r = (unsigned)wave_sum((int)wave_difference_num_upper(A, a, B, b, sk, up)); break;
      case GM_OP_DIFFERENCE_SET: r = (unsigned)wave_difference_set(A, a, B, b, sk, O); break;
      case GM_OP_DIFFERENCE_SET_UPPER: r = (unsigned)wave_difference_set_upper(A, a, B, b, sk, up, O); break;
      case GM_OP_COUNT_SMALLER: r = (unsigned)wave_sum((int)wave_count_smaller(up, A, a)); break;
      default: break;
    }
    if (lane == 0) out_num[i] = r;
  }
}

extern "C" int gm_setop_batch(int op, int64_t npairs, const int32_t *d_values, const int64_t *d_a_begin, const int64_t *d_a_end,
                              const int64_t *d_b_begin, const int64_t *d_b_end, const int32_t *d_upper, const int32_t *d_skip,
                              uint32_t *d_out_num, int32_t *d_out_values, void *stream) {
  if (op < 0 || op > GM_OP_COUNT_SMALLER || npairs < 0) return GM_ERR_INVALID;
  if (npairs == 0) return GM_OK;
  if (!d_values || !d_a_begin || !d_a_end || !d_b_begin || !d_b_end || !d_out_num) return GM_ERR_INVALID;
  const bool is_set = (op == GM_OP_INTERSECT_SET || op == GM_OP_INTERSECT_SET_UPPER || op == GM_OP_DIFFERENCE_SET ||
                       op == GM_OP_DIFFERENCE_SET_UPPER);
  if (is_set && !d_out_values) return GM_ERR_INVALID;
  const long long blocks = std::min<long long>((npairs + 3) / 4, 8192);
  hipLaunchKernelGGL(setop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, op, (long long)npairs, d_values,
                     (const long long *)d_a_begin, (const long long *)d_a_end, (const long long *)d_b_begin,
                     (const long long *)d_b_end, d_upper, d_skip, d_out_num, d_out_values);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// R-MAT key generator (tooling)
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long gm_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void rmat_kernel(int scale, long long n_edges, unsigned long long seed,
                                                   unsigned long long *__restrict__ keys) {
  const unsigned TA = 2448131358u;  // floor(0.57 * 2^32)
  const unsigned TB = 3264175144u;  // TA + floor(0.19 * 2^32)
  const unsigned TC = 4080218930u;  // TB + floor(0.19 * 2^32)
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_edges; i += stride) {
    const unsigned long long h = gm_mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
    unsigned long long s = 0, d = 0;
    for (int l = 0; l < scale; ++l) {
      const unsigned r = (unsigned)(gm_mix64(h + 0xD1B54A32D192ED03ull * (unsigned long long)(l + 1)) >> 32);
      const unsigned q = (r < TA) ? 0u : (r < TB) ? 1u : (r < TC) ? 2u : 3u;
      s = (s << 1) | (q >> 1);
      d = (d << 1) | (q & 1u);
    }
    if (s == d) {
      keys[2 * i] = ~0ull;
      keys[2 * i + 1] = ~0ull;
    } else {
      keys[2 * i] = (s << 32) | d;
      keys[2 * i + 1] = (d << 32) | s;
    }
  }
}

extern "C" int gm_rmat_keys(int scale, int64_t n_edges, uint64_t seed, uint64_t *d_keys, void *stream) {
  if (scale < 1 || scale > 30 || n_edges < 0 || !d_keys) return GM_ERR_INVALID;
  if (n_edges == 0) return GM_OK;
  const long long blocks = std::min<long long>((n_edges + 255) / 256, 65536);
  hipLaunchKernelGGL(rmat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scale, (long long)n_edges,
                     (unsigned long long)seed, (unsigned long long *)d_keys);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// PMC calibration stream: every lane reads one dword per iteration (the access width of the mining kernels'
// key loads); n*4 bytes are read exactly once, so FETCH_SIZE / (4n) gives the counter's scale for this width.
__global__ __launch_bounds__(256) void calib_stream_kernel(const int *__restrict__ buf, long long n, unsigned long long *out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += (unsigned)buf[i];
  acc = wave_sum_u64(acc);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

extern "C" int gm_calib_stream(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream) {
  if (!d_buf || !d_out || n < 0) return GM_ERR_INVALID;
  hipLaunchKernelGGL(calib_stream_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, d_buf, (long long)n,
                     (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// Stream ceiling: the fastest plain read this library can issue (16 B per lane, grid-stride, 8 workgroups per CU), used by
// bench.py as the MEASURED HBM ceiling next to the 8 TB/s spec (SURVEY.md 8d "Bounding roofline").
typedef int gm_v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_ceiling_kernel(const gm_v4i *__restrict__ buf, long long n4, unsigned long long *out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  gm_v4i acc = {0, 0, 0, 0};
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent 16-byte loads in flight per lane
    const gm_v4i a = __builtin_nontemporal_load(buf + i), b = __builtin_nontemporal_load(buf + i + stride);
    const gm_v4i c = __builtin_nontemporal_load(buf + i + 2 * stride), d = __builtin_nontemporal_load(buf + i + 3 * stride);
    acc ^= a ^ b ^ c ^ d;
  }
  for (; i < n4; i += stride) acc ^= buf[i];
  const unsigned long long s = wave_sum_u64((unsigned long long)(unsigned)(acc.x ^ acc.y ^ acc.z ^ acc.w));
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

extern "C" int gm_stream_ceiling(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream) {
  if (!d_buf || !d_out || n < 0 || ((uintptr_t)d_buf & 15)) return GM_ERR_INVALID;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  hipLaunchKernelGGL(stream_ceiling_kernel, dim3((unsigned)(cus * 8)), dim3(256), 0, (hipStream_t)stream, (const gm_v4i *)d_buf,
                     (long long)(n / 4), (unsigned long long *)d_out);
  HIP_TRY(hipGetLastError());
  return GM_OK;
}

// ------------------------------------------------------------------------------------------------
// wave-primitive self test
// ------------------------------------------------------------------------------------------------
__global__ void selftest_kernel(int *out) {
  __shared__ int lds[64];
  const int lane = threadIdx.x;
  const int x = (lane * 7 + 3) % 11;
  out[lane] = wave_incl_scan_add(x);
  const int y = (lane % 9 == 0) ? lane + 1 : 0;
  out[64 + lane] = wave_incl_scan_max(y);
  const unsigned long long m = __ballot((lane % 3) == 1);
  out[128 + lane] = rank_below(m);
  out[192 + lane] = lane_id();
  lds[lane] = lane * 3;  // ascending
  wave_sync();
  int pos;
  const bool f = contains(&lds[0], 64, lane * 2, &pos);
  out[256 + lane] = f ? pos : -1;
  out[320 + lane] = lower_bound(&lds[0], 64, lane * 2);
  out[384 + lane] = (int)wave_sum_u64((unsigned long long)lane + (1ull << 33));  // low word of 64*2^33 + 2016
  out[448 + lane] = (int)(wave_sum_u64((unsigned long long)lane + (1ull << 33)) >> 32);
}

extern "C" int gm_selftest(int device, int *n_fail) {
  if (n_fail) *n_fail = -1;
  HIP_TRY(hipSetDevice(device));
  int *d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(int) * 512));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  int h[512];
  hipError_t e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e, "selftest", __FILE__, __LINE__);
  int bad = 0, acc = 0, mx = 0, rk = 0;
  for (int l = 0; l < 64; ++l) {
    acc += (l * 7 + 3) % 11;
    bad += h[l] != acc;
    const int y = (l % 9 == 0) ? l + 1 : 0;
    mx = std::max(mx, y);
    bad += h[64 + l] != mx;
    bad += h[128 + l] != rk;
    rk += (l % 3) == 1;
    bad += h[192 + l] != l;
    const int key = l * 2;
    const int expect_pos = (key % 3 == 0 && key / 3 < 64) ? key / 3 : -1;
    bad += h[256 + l] != expect_pos;
    int lb = 0;
    while (lb < 64 && lb * 3 < key) ++lb;
    bad += h[320 + l] != lb;
    const unsigned long long tot = 64ull * (1ull << 33) + 2016ull;
    bad += h[384 + l] != (int)(unsigned)tot;
    bad += h[448 + l] != (int)(tot >> 32);
  }
  if (n_fail) *n_fail = bad;
  if (bad) { g_last_error = "wave primitive self test mismatch"; return GM_ERR_HIP; }
  return GM_OK;
}
