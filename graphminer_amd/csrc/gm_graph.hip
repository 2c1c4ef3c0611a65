// gm_graph.hip -- the device graph handle behind include/graphminer_amd.h: upload / adopt (GraphGPU::init, include/graph_gpu.h:69-122),
// Graph::orientation and Graph::sort_neighbors on the GPU (src/common/graph.cc:233-279,138-146), degree renumbering, download.
#include "gm_host.h"
#include <unordered_map>
#include "gm_scan.h"
#include "gm_setops.h"
using namespace gm;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_last_error;
thread_local TempPool *g_temp_pool = nullptr;

// ---- developer options (gm_mine.h; include/graphminer_amd.h gm_dev_option) ----
namespace {
std::mutex g_opt_mu;
std::list<std::pair<std::string, std::string>> g_opts;  // (a list: the strings gm_opt hands out stay where they are while other options change)
}  // namespace
const char *gm_opt(const char *name) {
  if (!name) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    for (const auto &kv : g_opts)
      if (kv.first == name) return kv.second.c_str();
  }
#ifdef GM_DEVEL
  return getenv(name);
#else
  return nullptr;
#endif
}
extern "C" const char *gm_dev_option_get(const char *name) { return gm_opt(name); }
extern "C" int gm_dev_option(const char *name, const char *value) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  if (!name) {
    g_opts.clear();
    return GM_OK;
  }
  if (!*name) return GM_ERR_INVALID;
  for (auto it = g_opts.begin(); it != g_opts.end(); ++it)
    if (it->first == name) {
      g_opts.erase(it);
      break;
    }
  if (value) g_opts.emplace_back(name, value);
  return GM_OK;
}

// ---- the cache of large temporaries (gm_host.h DevBuf) ----
namespace {
struct BigBlock { void *p; size_t bytes; int device; };
std::mutex g_big_mu;
std::vector<BigBlock> g_big_blocks;
constexpr size_t kBigCacheBudget = (size_t)16 << 30;  // per device
constexpr size_t kBigCacheBlocks = 512;               // per device (temporaries AND the persistent arrays of freed handles)
bool big_cache_on() { return gm_opt("GM_NO_TEMP_POOL") == nullptr; }  // (read at every call: tests switch it inside one process)
// Blocks of >= kBigCacheMinBytes that dev_malloc handed out (persistent arrays of a handle: key stream, task lists, matrices, tables): their
// sizes, so that dev_free can put them into the same per-device cache instead of giving them back to the driver -- a FRESH handle on a device
// that still has a live one then builds everything without a single hipMalloc / hipFree (round 6: both cost 0.01 - 0.2 ms on a good day and tens
// of ms per GB in windows of seconds, profiles/r05/alloc_jitter.txt; the driver's round-6 mid run saw first calls of 57 / 82 / 267 / 399 ms for
// 9 / 16 / 41 / 69).  The cache is emptied when the LAST handle of the device is freed, and whenever an allocation meets an out-of-memory error.
std::unordered_map<void *, size_t> g_dev_sizes;  // (under g_big_mu)
int g_live_handles[64] = {0};                    // (under g_big_mu) handles alive per device
bool cache_take(int dev, size_t bytes, size_t max_bytes, void **p, size_t *got) {  // (caller holds g_big_mu) the smallest cached block in [bytes, max_bytes]
  size_t best = g_big_blocks.size();
  for (size_t i = 0; i < g_big_blocks.size(); ++i) {
    const BigBlock &b = g_big_blocks[i];
    if (b.device == dev && b.bytes >= bytes && b.bytes <= max_bytes && (best == g_big_blocks.size() || b.bytes < g_big_blocks[best].bytes)) best = i;
  }
  if (best == g_big_blocks.size()) return false;
  *p = g_big_blocks[best].p;
  *got = g_big_blocks[best].bytes;
  g_big_blocks.erase(g_big_blocks.begin() + (long)best);
  return true;
}
}  // namespace
void dev_handle_born(int device) {
  std::lock_guard<std::mutex> lk(g_big_mu);
  if (device >= 0 && device < 64) ++g_live_handles[device];
}
bool dev_handle_died(int device) {  // true: it was the device's last one
  std::lock_guard<std::mutex> lk(g_big_mu);
  if (device < 0 || device >= 64) return true;
  return --g_live_handles[device] <= 0;
}
hipError_t dev_malloc_bytes(void **p, size_t bytes) {
  const bool cached = bytes >= kBigCacheMinBytes && big_cache_on();
  int dev = 0;
  if (cached) {
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_big_mu);
    size_t got = 0;
    if (cache_take(dev, bytes, bytes + bytes / 4, p, &got)) {  // (a persistent array: a block at most a quarter larger than asked for)
      g_dev_sizes[*p] = got;
      return hipSuccess;  // (big_cache_put synchronised the device before the block went in)
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();  // no room: the cached blocks of this device go back to the driver, then once more
    big_cache_trim();
    e = hipMalloc(p, bytes);
  }
  if (e == hipSuccess && cached) {
    std::lock_guard<std::mutex> lk(g_big_mu);
    g_dev_sizes[*p] = bytes;
  }
  return e;
}
void dev_free(void *p) {
  if (!p) return;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    auto it = g_dev_sizes.find(p);
    if (it != g_dev_sizes.end()) {
      bytes = it->second;
      g_dev_sizes.erase(it);
    }
  }
  if (bytes) big_cache_put(p, bytes);  // (synchronises the device; beyond the cache's budget: hipFree)
  else (void)hipFree(p);
}
hipError_t big_cache_get(void **p, size_t bytes, size_t *block_bytes) {
  *block_bytes = 0;
  const bool on = big_cache_on();
  if (on) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_big_mu);
    // (a temporary: the smallest block that holds the request without wasting more than 3 / 4 of itself)
    if (cache_take(dev, bytes, bytes * 4, p, block_bytes)) return hipSuccess;  // (big_cache_put synchronised the device before the block went in)
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    big_cache_trim();
    e = hipMalloc(p, bytes);
  }
  if (e == hipSuccess && on) *block_bytes = bytes;  // (0: a plain allocation, freed by hipFree)
  return e;
}
void big_cache_put(void *p, size_t block_bytes) {
  if (!p) return;
  (void)hipDeviceSynchronize();  // what hipFree did implicitly: nothing may still be using the block when it is handed out again
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (big_cache_on()) {
    std::lock_guard<std::mutex> lk(g_big_mu);
    size_t total = 0, blocks = 0;  // this device's share of the cache
    for (const BigBlock &b : g_big_blocks)
      if (b.device == dev) total += b.bytes, ++blocks;
    if (total + block_bytes <= kBigCacheBudget && blocks < kBigCacheBlocks) {
      g_big_blocks.push_back({p, block_bytes, dev});
      return;
    }
  }
  (void)hipFree(p);
}
void big_cache_trim() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::vector<BigBlock> mine;
  {
    std::lock_guard<std::mutex> lk(g_big_mu);
    for (size_t i = 0; i < g_big_blocks.size();) {
      if (g_big_blocks[i].device == dev) {
        mine.push_back(g_big_blocks[i]);
        g_big_blocks.erase(g_big_blocks.begin() + (long)i);
      } else ++i;
    }
  }
  for (auto &b : mine) (void)hipFree(b.p);
}

thread_local int g_last_hip_error = 0;
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  g_last_hip_error = (int)e;
  char buf[512];
  snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_last_error = buf;
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver) ? GM_ERR_NO_DEVICE
                                                                                                  : GM_ERR_HIP;
}
extern "C" const char *gm_strerror(int s) {
  switch (s) {
    case GM_OK: return "ok";
    case GM_ERR_INVALID: return "invalid argument";
    case GM_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
    case GM_ERR_HIP: return "HIP runtime error";
    case GM_ERR_TOO_LARGE: return "graph exceeds the 32-bit task index of the mining kernels (or the device memory)";
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

extern "C" int gm_graph_setup_times(const gm_graph *g, gm_setup_times *out) {
  if (!g || !out) return GM_ERR_INVALID;
  *out = g->setup;
  // cached derived handles (the oriented copy, renumbered copies -- and theirs) report through their owner
  for (const gm_graph *r : {g->dag_cache, g->relabel_cache[0], g->relabel_cache[1], g->relabel_cache[2]})
    if (r) {
      gm_setup_times t;
      const int rc = gm_graph_setup_times(r, &t);
      if (rc) return rc;
      out->orient_ms += t.orient_ms;
      out->table_ms += t.table_ms;
      out->bitmap_ms += t.bitmap_ms;
      out->relabel_ms += t.relabel_ms;
      out->other_ms += t.other_ms;
    }
  return GM_OK;
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
    if (b.d_bitmaps) dev_free(b.d_bitmaps);
    if (b.d_row_slot) dev_free(b.d_row_slot);
  }
  if (g->d_rp) dev_free(g->d_rp);
  if (g->d_rp64) dev_free(g->d_rp64);
  if (g->own_col && g->d_col) dev_free(g->d_col);
  if (g->d_edesc) dev_free(g->d_edesc);
  if (g->d_symdeg) dev_free(g->d_symdeg);
  if (g->d_trp) dev_free(g->d_trp);
  if (g->d_tdesc) dev_free(g->d_tdesc);
  if (g->d_tedge) dev_free(g->d_tedge);
  for (void *q : {(void *)g->d_emoff, (void *)g->d_tmoff, (void *)g->d_smask, (void *)g->d_sup_far_rows})
    if (q) dev_free(q);
  if (g->d_long_rows) dev_free(g->d_long_rows);
  if (g->d_long_prefix) dev_free(g->d_long_prefix);
  if (g->d_colk) dev_free(g->d_colk);
  if (g->d_tdesck) dev_free(g->d_tdesck);
  if (g->d_kst) dev_free(g->d_kst);
  for (void *q : {(void *)g->d_kst_et, (void *)g->d_tedgel, (void *)g->d_kst2, (void *)g->d_tdescl2})
    if (q) dev_free(q);
  if (g->d_kst_rp) dev_free(g->d_kst_rp);
  if (g->d_trpl) dev_free(g->d_trpl);
  if (g->d_tdescl) dev_free(g->d_tdescl);
  if (g->d_sup) dev_free(g->d_sup);
  free_clique_plans(g);
  if (g->d_wide_mat) dev_free(g->d_wide_mat);
  if (g->d_wide_sorted) dev_free(g->d_wide_sorted);
  if (g->d_wide_queue) dev_free(g->d_wide_queue);
  for (auto &st_ : g->aux_stream) if (st_) (void)hipStreamDestroy(st_);
  for (auto &ev_ : g->aux_done) if (ev_) (void)hipEventDestroy(ev_);
  if (g->pool.base) dev_free(g->pool.base);
  if (g->d_counters) dev_free(g->d_counters);
  if (g->d_scratch) dev_free(g->d_scratch);
  if (g->d_core) dev_free(g->d_core);
  for (void *q : {(void *)g->d_cg_tri, (void *)g->d_cg_rowbase, (void *)g->d_cg_bid, (void *)g->d_cg_blk})
    if (q) dev_free(q);
  if (g->d_csym) dev_free(g->d_csym);
  if (g->d_cfirst) dev_free(g->d_cfirst);
  if (g->d_idx0) dev_free(g->d_idx0);
  if (g->d_wblock_prefix) dev_free(g->d_wblock_prefix);
  if (g->d_house_prefix) dev_free(g->d_house_prefix);
  if (g->d_pent_touched) dev_free(g->d_pent_touched);
  if (g->d_house_t) dev_free(g->d_house_t);
  if (g->d_house_tlt) dev_free(g->d_house_tlt);
  if (g->d_house_tasks) dev_free(g->d_house_tasks);
  if (g->d_house_acc) dev_free(g->d_house_acc);
  if (g->d_house_touched) dev_free(g->d_house_touched);
  if (g->d_house_bnd) dev_free(g->d_house_bnd);
  if (g->d_house_lds_tasks) dev_free(g->d_house_lds_tasks);
  if (g->d_house_cut_tasks) dev_free(g->d_house_cut_tasks);
  if (g->d_rect_tasks) dev_free(g->d_rect_tasks);
  if (g->d_rect_bnd) dev_free(g->d_rect_bnd);
  if (g->d_rect_lds_tasks) dev_free(g->d_rect_lds_tasks);
  if (g->d_rect_cut_tasks) dev_free(g->d_rect_cut_tasks);
  if (g->d_rect_acc) dev_free(g->d_rect_acc);
  for (auto &pr : g->ev)
    for (auto &e : pr)
      if (e) (void)hipEventDestroy(e);
  const bool last = g->counted && dev_handle_died(g->device);
  delete g;
  // the blocks kept for reuse -- large temporaries, and the persistent arrays of freed handles (dev_free) -- go back to the driver with the
  // device's LAST handle (not inside a timed call); while another handle lives they serve the next one
  if (last) big_cache_trim();
}

int finish_handle(gm_graph *g) {
  if (!g->counted) {
    g->counted = true;
    dev_handle_born(g->device);
  }
  static std::once_flag warm;  // once per process: load every kernel module now, not inside the first (timed) table build or launch
  std::call_once(warm, [] {
    gm_touch_mine();
    gm_touch_mine_wide();
    gm_touch_hrow();
    gm_touch_tch();
    gm_touch_sup();
    gm_touch_cbuild();
    gm_touch_cmma();
    gm_touch_cgather();
    gm_touch_ctc();
    gm_touch_sgl();
    gm_touch_tables();
    gm_touch_launch();
    gm_touch_tools();
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
  });
  HIP_TRY(dev_malloc(&g->d_counters, 64));
  HIP_TRY(hipMemset(g->d_counters, 0, 64));
  for (auto &pr : g->ev)
    for (auto &e : pr) HIP_TRY(hipEventCreate(&e));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, g->device));
  g->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return GM_OK;  // (max_deg: set by the caller from the device-side pass that produced the offsets)
}

// host copy of the 32-bit offsets, fetched from the device on first use
int host_rp(gm_graph *g, const std::vector<int> **out) {
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

// validation of 64-bit offsets that stay 64-bit: err bit 0 = not an offset array, bit 1 = a row of 2^31 entries or more; info[1] = longest row
__global__ __launch_bounds__(256) void check_offsets64_kernel(const long long *__restrict__ rp64, int nv, long long ne, int *__restrict__ info) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  int err = 0, md = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v <= nv; v += stride) {
    const long long x = rp64[v];
    if (v == 0 && x != 0) err |= 1;
    if (v == nv && x != ne) err |= 1;
    if (v > 0) {
      const long long d = x - rp64[v - 1];
      if (d < 0) err |= 1;
      else if (d >= 0x7fffffffLL) err |= 2;
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
  DevBuf<int> info;
  HIP_TRY(info.alloc(2));
  HIP_TRY(hipMemset(info.p, 0, 8));
  const long long blocks = std::min<long long>(((long long)g->nv + 256) / 256, 4096);
  long long big_from = 0x7fffffffLL;
  if (const char *e = gm_opt("GM_BIG_NE")) big_from = std::max(0ll, atoll(e));  // (tests: the 64-bit paths on small graphs)
  if (g->ne >= big_from) {  // a big handle: its own copy of the 64-bit offsets
    HIP_TRY(dev_malloc(&g->d_rp64, sizeof(long long) * ((size_t)g->nv + 1)));
    HIP_TRY(hipMemcpy(g->d_rp64, d_rp64, sizeof(long long) * ((size_t)g->nv + 1), hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(check_offsets64_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, (const long long *)g->d_rp64, g->nv, g->ne, info.p);
    HIP_TRY(hipGetLastError());
    int h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, info.p, 8, hipMemcpyDeviceToHost));
    if (h[0] & 1) return GM_ERR_FORMAT;
    if (h[0] & 2) return GM_ERR_TOO_LARGE;
    g->max_deg = h[1];
    return GM_OK;
  }
  HIP_TRY(dev_malloc(&g->d_rp, sizeof(int) * ((size_t)g->nv + 1)));
  hipLaunchKernelGGL(convert_offsets_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, (const long long *)d_rp64, g->nv, g->ne, g->d_rp, info.p);
  HIP_TRY(hipGetLastError());  // (a failed launch would leave info == 0 and an uninitialised d_rp behind a passing validation)
  int h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, info.p, 8, hipMemcpyDeviceToHost));
  if (h[0] & 1) return GM_ERR_FORMAT;
  if (h[0] & 2) return GM_ERR_TOO_LARGE;
  g->max_deg = h[1];
  return GM_OK;
}

// ne >= 2^31: a BIG handle (d_rp64 instead of d_rp). It can be oriented (the DAG of twitter40 / friendster has < 2^31 entries), its
// triangles / wedges counted through the formula solver (gm_motif, k = 3) and downloaded; the mining kernels that walk the symmetric
// graph itself index it with 32 bits and refuse it with GM_ERR_TOO_LARGE.
static int check_sizes(long long nv, long long ne) {
  if (nv < 0 || ne < 0) return GM_ERR_INVALID;
  if (nv >= 0x7ffffffeLL || ne >= (1ll << 40)) return GM_ERR_TOO_LARGE;
  return GM_OK;
}


int convert_offsets(const int64_t *rp64, int nv, long long ne, std::vector<int> &out) {
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
  if ((e = dev_malloc(&g->d_col, sizeof(int) * (size_t)std::max<long long>(g->ne, 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc(col)", __FILE__, __LINE__));
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
  if (g->d_rp64) { g_last_error = "gm_graph_sort_neighbors: not for graphs of 2^31 entries or more (the segmented sort counts items in 32 bits)"; return GM_ERR_TOO_LARGE; }
  // before any solver ran: every cached structure describes the rows as they are now (tables, descriptors, task lists, derived
  // handles, the per-pattern tables of the SgL / 4-motif paths, hub bitmaps, the sum of C(d,2))
  if (!g->tables.empty() || g->d_edesc || g->d_trp || g->d_tdesc || g->d_kst_rp || g->dag_cache || g->relabel_cache[0] || g->relabel_cache[1] || g->d_idx0 ||
      g->d_wblock_prefix || g->d_rect_tasks || g->d_house_t || g->d_house_tlt || g->d_house_tasks || g->d_house_prefix || !g->bitmap_sets.empty() ||
      g->wide_valid || g->sum_c2_valid)
    return GM_ERR_INVALID;
  HIP_TRY(hipDeviceSynchronize());  // a caller stream may still be reading a borrowed col_idx array: the sort runs on the null stream
  HIP_TRY(hipSetDevice(g->device));
  PoolScope pool(g);
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
  g->sorted_state = 0;  // (checked again by the first solver: duplicates are still not "strictly ascending")
  g->topo_state = 0;
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
  if (row_ptr && g->d_rp64) {
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipMemcpy(row_ptr, g->d_rp64, sizeof(int64_t) * ((size_t)g->nv + 1), hipMemcpyDeviceToHost));
  } else if (row_ptr) {
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

// OffT: the offset type of the SYMMETRIC graph that is read -- int, or long long for a graph of 2^31 entries or more (twitter40,
// friendster: src/triangle/README.md:60-61); the oriented graph that is written has 32-bit offsets either way.
template <class OffT>
struct OrientSegT {
  int row;
  OffT begin, end;  // CSR entry range [begin,end) of `row`
  int out;          // output offset of the segment (pass 1)
};

template <class OffT>
__global__ __launch_bounds__(256) void orient_short_kernel(int nv, const OffT *__restrict__ rp, const int *__restrict__ col,
                                                           const int *__restrict__ sdeg /* symmetric degrees: one random load per entry instead of two offsets */,
                                                           int *__restrict__ new_deg, const int *__restrict__ new_rp,
                                                           int *__restrict__ new_col, int pass, int *__restrict__ tmp_col) {
  constexpr int G = 8, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G, gl = lane % G;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = wave * RPW; s0 < nv; s0 += nwaves * RPW) {
    const int s = s0 + grp;
    OffT b = 0;
    int ds = 0;
    if (s < nv) { b = rp[s]; ds = (int)(rp[s + 1] - b); }
    const int full = ds;
    if (ds > kOrientShort) ds = 0;  // long rows belong to the segment kernel
    const int maxds = wave_max_nonneg(ds);
    int n = 0;
    const int ob = (pass && ds > 0) ? new_rp[s] : 0;
    // every entry of the (<= 64-entry) rows is requested before the first is looked at, then every degree: two memory latencies per
    // trip instead of two per eight entries (the passes were latency bound: 94 G entries/s whatever the gather touched)
    constexpr int KS = kOrientShort / G;
    int xd[KS], xg[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      xd[k] = -1;
      if (k * G < maxds && k * G + gl < ds) xd[k] = col[b + k * G + gl];
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      xg[k] = 0;
      if (k * G < maxds && xd[k] >= 0) xg[k] = sdeg[xd[k]];
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (k * G < maxds) {
        const int d = xd[k];
        const bool keep = d >= 0 && dag_keep(full, s, xg[k], d);
        const unsigned long long m = (__ballot(keep) >> (grp * G)) & 0xffull;
        if (keep) {
          const int at = n + __popcll(m & ((1ull << gl) - 1ull));
          if (pass) new_col[ob + at] = d;
          else if (tmp_col) tmp_col[b + at] = d;  // kept entries packed at the row's own place: pass 1 is then a copy, not a second gather
        }
        n += __popcll(m);
      }
    }
    if (!pass && gl == 0 && s < nv && full <= kOrientShort) new_deg[s] = n;
  }
}

// one wave per segment; pass 0 adds the segment's count to its row's new degree (and keeps it per segment for the offsets)
template <class OffT>
__global__ __launch_bounds__(256) void orient_seg_kernel(int nseg, const OrientSegT<OffT> *__restrict__ segs, const OffT *__restrict__ rp,
                                                         const int *__restrict__ col, const int *__restrict__ sdeg, int *__restrict__ seg_count,
                                                         int *__restrict__ new_deg, int *__restrict__ new_col, int pass, int *__restrict__ tmp_col) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int sg = wave; sg < nseg; sg += nwaves) {
    const OrientSegT<OffT> q = segs[sg];
    const int ds = (int)(rp[q.row + 1] - rp[q.row]);
    int n = 0;
    for (OffT base = q.begin; base < q.end; base += 256) {  // four 64-entry steps requested together
      int xd[4], xg[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const OffT i = base + k * 64 + lane;
        xd[k] = i < q.end ? col[i] : -1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) xg[k] = xd[k] >= 0 ? sdeg[xd[k]] : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int d = xd[k];
        const bool keep = d >= 0 && dag_keep(ds, q.row, xg[k], d);
        const unsigned long long m = __ballot(keep);
        if (keep) {
          if (pass) new_col[q.out + n + rank_below(m)] = d;
          else if (tmp_col) tmp_col[q.begin + n + rank_below(m)] = d;
        }
        n += __popcll(m);
      }
    }
    if (!pass && lane == 0) {
      seg_count[sg] = n;
      if (n) atomicAdd(&new_deg[q.row], n);
    }
  }
}

// segment table of the long rows, built on the device: seg_first[v] = exclusive scan of ceil(d/kOrientSeg) over the long rows
template <class OffT>
__global__ __launch_bounds__(256) void orient_segcount_kernel(int nv, const OffT *__restrict__ rp, int *__restrict__ nseg_of) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v > nv) return;
  int n = 0;
  if (v < nv) {
    const int d = (int)(rp[v + 1] - rp[v]);
    n = d > kOrientShort ? (d + kOrientSeg - 1) / kOrientSeg : 0;
  }
  nseg_of[v] = n;  // (nseg_of[nv] = 0: the scan's last element is the total)
}
template <class OffT>
__global__ __launch_bounds__(256) void orient_segfill_kernel(int nv, const OffT *__restrict__ rp, const int *__restrict__ seg_first,
                                                             OrientSegT<OffT> *__restrict__ segs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const OffT b = rp[v], e = rp[v + 1];
  if (e - b <= kOrientShort) return;
  int k = seg_first[v];
  for (OffT s0 = b; s0 < e; s0 += kOrientSeg) segs[k++] = {v, s0, (s0 + kOrientSeg < e) ? (OffT)(s0 + kOrientSeg) : e, 0};
}
// output offset of every segment: the row's new offset + the kept entries of the row's earlier segments (one thread per long row)
template <class OffT>
__global__ __launch_bounds__(256) void orient_segout_kernel(int nv, const OffT *__restrict__ rp, const int *__restrict__ seg_first,
                                                            const int *__restrict__ seg_count, const int *__restrict__ new_rp,
                                                            OrientSegT<OffT> *__restrict__ segs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  if (rp[v + 1] - rp[v] <= kOrientShort) return;
  int run = new_rp[v];
  for (int k = seg_first[v]; k < seg_first[v + 1]; ++k) {
    segs[k].out = run;
    run += seg_count[k];
  }
}
// pass 1 as a copy: the kept entries of a short row sit packed at the row's old offset (pass 0), eight lanes move them to the new one
template <class OffT>
__global__ __launch_bounds__(256) void orient_copy_short_kernel(int nv, const OffT *__restrict__ rp, const int *__restrict__ new_rp,
                                                                const int *__restrict__ tmp_col, int *__restrict__ new_col) {
  constexpr int G = 8, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int grp = lane / G, gl = lane % G;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int s0 = wave * RPW; s0 < nv; s0 += nwaves * RPW) {
    const int s = s0 + grp;
    if (s >= nv) continue;
    const OffT b = rp[s];
    if (rp[s + 1] - b > kOrientShort) continue;
    const int ob = new_rp[s], n = new_rp[s + 1] - ob;
    for (int i = gl; i < n; i += G) new_col[ob + i] = tmp_col[b + i];
  }
}
template <class OffT>
__global__ __launch_bounds__(256) void orient_copy_seg_kernel(int nseg, const OrientSegT<OffT> *__restrict__ segs, const int *__restrict__ seg_count,
                                                              const int *__restrict__ tmp_col, int *__restrict__ new_col) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int sg = wave; sg < nseg; sg += nwaves) {
    const OrientSegT<OffT> q = segs[sg];
    const int n = seg_count[sg];
    for (int i = lane; i < n; i += 64) new_col[q.out + i] = tmp_col[q.begin + i];
  }
}
__global__ __launch_bounds__(256) void max_degree_kernel(int nv, const int *__restrict__ rp, int *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  int md = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) md = max(md, rp[v + 1] - rp[v]);
  md = gm::wave_max_nonneg(md);
  if ((threadIdx.x & 63) == 0 && md) atomicMax(out, md);
}

template <class OffT>
__global__ __launch_bounds__(256) void symdeg_kernel(int nv, const OffT *__restrict__ rp, int *__restrict__ deg) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nv) deg[v] = (int)(rp[v + 1] - rp[v]);
}
// new offsets of an orientation whose input has 64-bit offsets: the scan runs in 64 bits (the result must be checked against the
// 32-bit limit of the oriented handle before it is narrowed)
__global__ __launch_bounds__(256) void narrow_offsets_kernel(int nv, const long long *__restrict__ in, int *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v <= nv) out[v] = (int)in[v];
}

template <class OffT>
static int orient_impl(const gm_graph *sym, const OffT *rp_in, gm_graph **out) {
  HIP_TRY(hipSetDevice(sym->device));
  SetupTimer timer;
  PoolScope pool(const_cast<gm_graph *>(sym));
  setup_trace("orient: begin");
  const int nv = sym->nv;
  const unsigned vb = (unsigned)((nv + 256) / 256);  // blocks covering v = 0 .. nv
  ScanTemp tmp;
  // the symmetric degrees first: the passes compare (degree, id) of both endpoints of every entry; the oriented handle keeps them (what
  // the topological renumbering of this DAG sorts by -- get_relabeled mode 2)
  int *sdeg = nullptr;
  HIP_TRY(dev_malloc(&sdeg, sizeof(int) * (size_t)std::max(nv, 1)));
  struct SdegGuard { int *&p; ~SdegGuard() { if (p) dev_free(p); } } sdeg_guard{sdeg};
  if (nv > 0) hipLaunchKernelGGL((symdeg_kernel<OffT>), dim3(vb), dim3(256), 0, 0, nv, rp_in, sdeg);
  // segment table of the long rows (device): counts -> exclusive scan -> fill
  DevBuf<int> nseg_of, seg_first, deg, segcnt;
  DevBuf<OrientSegT<OffT>> segs;
  HIP_TRY(nseg_of.alloc((size_t)nv + 1));
  HIP_TRY(seg_first.alloc((size_t)nv + 1));
  hipLaunchKernelGGL((orient_segcount_kernel<OffT>), dim3(vb), dim3(256), 0, 0, nv, rp_in, nseg_of.p);
  HIP_TRY(dev_exclusive_sum(tmp, nseg_of.p, seg_first.p, (size_t)nv + 1));
  int nseg = 0;
  HIP_TRY(hipMemcpy(&nseg, seg_first.p + nv, sizeof(int), hipMemcpyDeviceToHost));
  HIP_TRY(segs.alloc((size_t)nseg));
  HIP_TRY(segcnt.alloc((size_t)nseg));
  if (nseg) hipLaunchKernelGGL((orient_segfill_kernel<OffT>), dim3(vb), dim3(256), 0, 0, nv, rp_in, seg_first.p, segs.p);
  // pass 0: new degrees (short rows write, segments of long rows add)
  HIP_TRY(deg.alloc((size_t)nv + 1));
  HIP_TRY(hipMemsetAsync(deg.p, 0, sizeof(int) * ((size_t)nv + 1), 0));
  const int bs = std::max(1, std::min((nv + 31) / 32, sym->cu_count * 8));
  const int bl = std::max(1, std::min((nseg + 3) / 4, sym->cu_count * 8));
  // (the kept entries packed in a scratch copy at their rows' old offsets: pass 1 copies instead of gathering the degrees again --
  //  the gathers are the cost of a pass, one 64-byte sector per entry; GM_ORIENT_TWO_GATHERS=1 keeps round 3's second gather pass)
  const char *env_two = gm_opt("GM_ORIENT_TWO_GATHERS");
  const bool two_gathers = env_two && *env_two == '1';
  DevBuf<int> packed;
  if (!two_gathers && sym->ne > 0 && packed.alloc((size_t)sym->ne) != hipSuccess) { (void)hipGetLastError(); packed.p = nullptr; }  // (no room: gather twice)
  hipLaunchKernelGGL((orient_short_kernel<OffT>), dim3(bs), dim3(256), 0, 0, nv, rp_in, sym->d_col, sdeg, deg.p, (const int *)nullptr, (int *)nullptr, 0, packed.p);
  if (nseg)
    hipLaunchKernelGGL((orient_seg_kernel<OffT>), dim3(bl), dim3(256), 0, 0, nseg, segs.p, rp_in, sym->d_col, sdeg, segcnt.p, deg.p, (int *)nullptr, 0, packed.p);
  setup_trace("orient: segments + degrees");
  // new offsets = exclusive scan of the new degrees (parallel_prefix_sum, include/scan.h:5-35)
  gm_graph *g = new gm_graph();
  g->device = sym->device;
  g->nv = nv;
  auto fail = [&](int code) { gm_graph_free(g); return code; };
  hipError_t e;
  if ((e = dev_malloc(&g->d_rp, sizeof(int) * ((size_t)nv + 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  if constexpr (sizeof(OffT) == 8) {
    // the oriented graph of a symmetric graph of >= 2^31 entries: half of them -- checked in 64 bits before the offsets are narrowed
    DevBuf<long long> rp64;
    if ((e = rp64.alloc((size_t)nv + 1)) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
    if ((e = dev_exclusive_sum(tmp, deg.p, rp64.p, (size_t)nv + 1)) != hipSuccess) return fail(hip_fail(e, "ExclusiveSum", __FILE__, __LINE__));
    long long total = 0;
    if ((e = hipMemcpy(&total, rp64.p + nv, sizeof(long long), hipMemcpyDeviceToHost)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy", __FILE__, __LINE__));
    if (total >= 0x7fffffffLL) {
      g_last_error = "gm_graph_orient: the oriented graph has " + std::to_string(total) + " entries (>= 2^31: beyond the 32-bit task index of the mining kernels)";
      return fail(GM_ERR_TOO_LARGE);
    }
    hipLaunchKernelGGL(narrow_offsets_kernel, dim3(vb), dim3(256), 0, 0, nv, rp64.p, g->d_rp);
  } else {
    if ((e = dev_exclusive_sum(tmp, deg.p, g->d_rp, (size_t)nv + 1)) != hipSuccess) return fail(hip_fail(e, "ExclusiveSum", __FILE__, __LINE__));
  }
  DevBuf<int> md;
  if ((e = md.alloc(1)) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  (void)hipMemsetAsync(md.p, 0, sizeof(int), 0);
  hipLaunchKernelGGL(max_degree_kernel, dim3((unsigned)std::min<long long>(((long long)nv + 255) / 256, 2048)), dim3(256), 0, 0, nv, g->d_rp, md.p);
  int ne_new = 0, max_deg = 0;
  if ((e = hipMemcpy(&ne_new, g->d_rp + nv, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy", __FILE__, __LINE__));
  if ((e = hipMemcpy(&max_deg, md.p, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(hip_fail(e, "hipMemcpy", __FILE__, __LINE__));
  g->ne = ne_new;
  g->max_deg = max_deg;
  if ((e = dev_malloc(&g->d_col, sizeof(int) * (size_t)std::max(ne_new, 1))) != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
  setup_trace("orient: scan, max degree, allocations");
  // pass 1: compact
  if (nseg) hipLaunchKernelGGL((orient_segout_kernel<OffT>), dim3(vb), dim3(256), 0, 0, nv, rp_in, seg_first.p, segcnt.p, g->d_rp, segs.p);
  if (packed.p) {
    hipLaunchKernelGGL((orient_copy_short_kernel<OffT>), dim3(bs), dim3(256), 0, 0, nv, rp_in, g->d_rp, packed.p, g->d_col);
    if (nseg) hipLaunchKernelGGL((orient_copy_seg_kernel<OffT>), dim3(bl), dim3(256), 0, 0, nseg, segs.p, segcnt.p, packed.p, g->d_col);
  } else {
    hipLaunchKernelGGL((orient_short_kernel<OffT>), dim3(bs), dim3(256), 0, 0, nv, rp_in, sym->d_col, sdeg, (int *)nullptr, g->d_rp, g->d_col, 1, (int *)nullptr);
    if (nseg)
      hipLaunchKernelGGL((orient_seg_kernel<OffT>), dim3(bl), dim3(256), 0, 0, nseg, segs.p, rp_in, sym->d_col, sdeg, (int *)nullptr, (int *)nullptr, g->d_col, 1, (int *)nullptr);
  }
  if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) return fail(hip_fail(e, "orient kernels", __FILE__, __LINE__));
  setup_trace("orient: compact");
  int rc = finish_handle(g);
  if (rc) { gm_graph_free(g); return rc; }
  g->d_symdeg = sdeg;
  sdeg = nullptr;
  setup_trace("orient: finish_handle + degrees");
  g->setup.orient_ms = timer.ms();
  *out = g;
  return GM_OK;
}

// the renumbered copy a solver would run on (get_relabeled), borrowed: for tests and tools that want to look at it
extern "C" int gm_graph_renumbered(gm_graph *g, int mode, gm_graph **view) {
  if (!g || !view || mode < 0 || mode > 2) return GM_ERR_INVALID;
  *view = nullptr;
  if (g->d_rp64) return GM_ERR_TOO_LARGE;
  // a row with a DUPLICATE neighbour has no renumbered row (an entry's place is its rank among the row's new ids: two equal ids take one
  // slot and leave another unwritten); such a handle is refused here as the solvers refuse it
  bool sorted = false;
  const int rc = graph_rows_sorted(g, &sorted);
  if (rc) return rc;
  if (!sorted) {
    g_last_error = "gm_graph_renumbered: the rows are not strictly ascending (unsorted or duplicate neighbours); gm_graph_sort_neighbors sorts, duplicates must be removed by the caller";
    return GM_ERR_INVALID;
  }
  const int rc2 = get_relabeled(g, mode, view);
  if (rc2 == GM_OK && *view && (*view)->sorted_state == 2) {
    *view = nullptr;
    g_last_error = "gm_graph_renumbered: the renumbered copy has duplicate neighbours";
    return GM_ERR_INVALID;
  }
  return rc2;
}

extern "C" int gm_graph_orient(const gm_graph *sym, gm_graph **out) {
  if (!sym || !out) return GM_ERR_INVALID;
  *out = nullptr;
  return sym->d_rp64 ? orient_impl<long long>(sym, sym->d_rp64, out) : orient_impl<int>(sym, sym->d_rp, out);
}

// ------------------------------------------------------------------------------------------------
// Renumbering. A pattern count does not depend on the vertex numbering, but the work of some kernels does:
//  * the SgL kernels anchor a match at its largest vertex id and walk smaller ids. With ids ascending in degree the 2-path walks of
//    rectangle and the (v0, v1 < v0, v3) tasks of house go through low-degree vertices (R-MAT-16: 522 M -> 128 M 2-paths, 898 M ->
//    163 M tasks); with ids descending in degree the wedges (v0; v2 < v1 < v0) of pentagon avoid the hubs (99 M -> 33 M);
//  * the 4-clique kernels want a TOPOLOGICAL numbering of the DAG (every edge from a smaller to a larger id): the adjacency matrix of
//    N+(u) is then strictly upper triangular -- the in-edge tasks of gm_cbuild.hip stream half a list, the pair counts skip the words
//    below the diagonal. Ids ascending in (in-degree + out-degree, old id) are topological for every DAG that Graph::orientation
//    produced (src/common/graph.cc:246-247 keeps s -> d iff (deg, id) grows); any other DAG is checked and left as it is.
// The copy is built once per handle, on the device: one 64-bit key (degree, id) per vertex and one (new row, new neighbour) per CSR
// entry, two hipCUB radix sorts -- the rows come out ascending. (Round 2 sorted the vertices on the host.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relabel_indeg_kernel(long long ne, const int *__restrict__ col, int *__restrict__ indeg) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) atomicAdd(&indeg[col[e]], 1);
}
// (deg: the sorting degree itself when given -- the symmetric degrees a DAG remembers from its orientation; else out-degree + indeg)
__global__ __launch_bounds__(256) void relabel_vkeys_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ indeg, const int *__restrict__ deg,
                                                            unsigned long long *__restrict__ keys) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const unsigned long long d = deg ? (unsigned long long)deg[v] : (unsigned long long)(rp[v + 1] - rp[v]) + (indeg ? (unsigned long long)indeg[v] : 0ull);
  keys[v] = (d << 32) | (unsigned long long)(unsigned)v;
}
// position i of the sorted (degree, id) keys -> new id (ascending: i; descending: nv - 1 - i); the new row's length rides along
__global__ __launch_bounds__(256) void relabel_newid_kernel(int nv, const unsigned long long *__restrict__ sorted, int descending,
                                                            const int *__restrict__ rp, int *__restrict__ newid, int *__restrict__ newdeg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > nv) return;
  if (i == nv) { newdeg[nv] = 0; return; }
  const int v = (int)(unsigned)(sorted[i] & 0xffffffffull);
  const int id = descending ? nv - 1 - i : i;
  newid[v] = id;
  newdeg[id] = rp[v + 1] - rp[v];
}
// one key (new row << bits | new neighbour) per entry; a thread takes four consecutive entries: one bisection of the offsets for the
// first, a short walk for the others (a bisection per entry: 0.75 of the 3.5 ms a renumbering of R-MAT-22 took)
__global__ __launch_bounds__(256) void relabel_keys_kernel(int nv, long long ne, const int *__restrict__ rp, const int *__restrict__ col,
                                                           const int *__restrict__ newid, int bits, unsigned long long *__restrict__ keys) {
  const long long e0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e0 >= ne) return;
  int lo = 0, hi = nv - 1;  // row of entry e0: the largest u with rp[u] <= e0
  while (lo < hi) {
    const int mid = (int)(((long long)lo + hi + 1) >> 1);
    if (rp[mid] <= e0) lo = mid; else hi = mid - 1;
  }
  int next = rp[lo + 1];
  unsigned long long row = (unsigned long long)(unsigned)newid[lo] << bits;
  const int n = (int)min(4ll, ne - e0);
  for (int k = 0; k < n; ++k) {
    const long long e = e0 + k;
    while (e >= next) {  // (empty rows in between)
      ++lo;
      next = rp[lo + 1];
      row = (unsigned long long)(unsigned)newid[lo] << bits;
    }
    keys[e] = row | (unsigned long long)(unsigned)newid[col[e]];
  }
}

// The rows of the copy WITHOUT a device-wide sort (round 4): the new rows are walked in their new order (oldid = the sorted vertex keys),
// so the writes are coalesced and the eight rows of a wave have neighbouring degrees.
//  * a row of <= kRelabelShort entries: eight lanes gather the new ids of its entries into LDS and every entry's place is its RANK, the
//    number of entries of the row below it (n LDS broadcasts against <= 8 values per lane); longer rows are LISTED by this kernel;
//  * a listed row of <= kRelabelMid entries: one wave, bitonic network in LDS over the next power of two (relabel_rows_long_kernel);
//  * a listed row of <= kRelabelLdsMax entries: the same network by a whole workgroup (relabel_rows_block_kernel);
//  * a row beyond that (the hubs of a symmetric graph): new ids written unsorted, one segmented radix sort over those rows' segments
//    (relabel_rows_huge_kernel + get_relabeled).
// Two equal entries of a row (a duplicate in the input) set *dup, like the sorted keys did.
constexpr int kRelabelShort = 64;
constexpr int kRelabelMid = 1024;
constexpr int kRelabelLdsMax = 4096;

__global__ __launch_bounds__(256) void relabel_rows_short_kernel(int nv, const unsigned long long *__restrict__ vsorted, int descending,
                                                                 const int *__restrict__ rp, const int *__restrict__ col,
                                                                 const int *__restrict__ newid, const int *__restrict__ new_rp,
                                                                 int *__restrict__ new_col, int *__restrict__ dup,
                                                                 int *__restrict__ long_rows /* [cap]: rows of 65 .. kRelabelMid from the front, longer from the back */,
                                                                 int cap, int *__restrict__ long_count /* [3] */,
                                                                 int *__restrict__ huge_rows /* rows beyond kRelabelLdsMax entries */) {
  constexpr int G = 8, RPW = 64 / G, K = kRelabelShort / G;
  __shared__ int vals[4][RPW][kRelabelShort];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int grp = lane / G, gl = lane % G;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  int *mine = vals[wib][grp];
  for (int r0 = wave * RPW; r0 < nv; r0 += nwaves * RPW) {
    const int r = r0 + grp;  // new row
    int n = 0, b = 0, ob = 0;
    if (r < nv) {
      const int v = (int)(unsigned)(vsorted[descending ? nv - 1 - r : r] & 0xffffffffull);
      b = rp[v];
      n = rp[v + 1] - b;
      ob = new_rp[r];
      if (n > kRelabelShort) {  // the long-row kernels': listed (in the order the waves get here: neighbours in degree spread over the list)
        if (gl == 0) {
          if (n <= kRelabelMid) long_rows[atomicAdd(&long_count[0], 1)] = r;
          else if (n <= kRelabelLdsMax) long_rows[cap - 1 - atomicAdd(&long_count[1], 1)] = r;
          else huge_rows[atomicAdd(&long_count[2], 1)] = r;
        }
        n = 0;
      }
    }
    const int maxn = wave_max_nonneg(n);
    if (maxn == 0) continue;
    const int steps = (maxn + G - 1) / G;
    int x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      x[k] = 0x7fffffff;
      if (k < steps) {
        const int i = k * G + gl;
        if (i < n) {
          x[k] = newid[col[b + i]];
          mine[i] = x[k];
        }
      }
    }
    wave_sync();
    int below[K];
#pragma unroll
    for (int k = 0; k < K; ++k) below[k] = 0;
    bool same = false;
    for (int j = 0; j < maxn; ++j) {
      if (j < n) {
        const int y = mine[j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (k < steps) {
            below[k] += (y < x[k]) ? 1 : 0;
            same |= (y == x[k]) && (j != k * G + gl);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k < steps && k * G + gl < n) new_col[ob + below[k]] = x[k];
    if (same) *dup = 1;
    wave_sync();
  }
}

// one wave per listed row of 65 .. CAP = kRelabelMid entries, the last listed (longest) first; CAP entries of LDS per wave
template <int CAP>
__global__ __launch_bounds__(256) void relabel_rows_long_kernel(int nv, const unsigned long long *__restrict__ vsorted, int descending,
                                                                const int *__restrict__ rp, const int *__restrict__ col,
                                                                const int *__restrict__ newid, const int *__restrict__ new_rp,
                                                                int *__restrict__ new_col, int *__restrict__ dup,
                                                                const int *__restrict__ long_rows, int cap, const int *__restrict__ long_count) {
  __shared__ int lds_rows[4 * CAP];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  int *a = lds_rows + wib * CAP;
  const int count = long_count[CAP == kRelabelMid ? 0 : 1];
  for (int q = wave; q < count; q += nwaves) {
    const int r = CAP == kRelabelMid ? long_rows[count - 1 - q] : long_rows[cap - count + q];
    const int v = (int)(unsigned)(vsorted[descending ? nv - 1 - r : r] & 0xffffffffull);
    const int b = rp[v], n = rp[v + 1] - b, ob = new_rp[r];
    int P = 128;
    while (P < n) P <<= 1;
    for (int i = lane; i < P; i += 64) a[i] = i < n ? newid[col[b + i]] : 0x7fffffff;
    wave_sync();
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        // compare-exchange t works on the pair (i, i | j); two of them per lane and step so that four LDS reads are in flight
        for (int t = lane; t < half; t += 128) {
          const int t2 = t + 64;
          const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1)), i1 = ((t2 & ~(j - 1)) << 1) | (t2 & (j - 1));
          const int lo0 = a[i0], hi0 = a[i0 | j];
          const bool two = t2 < half;  // (wave-uniform: only the 128-entry network has a single round of 64 exchanges)
          int lo1 = 0, hi1 = 0;
          if (two) { lo1 = a[i1]; hi1 = a[i1 | j]; }
          if ((lo0 > hi0) == ((i0 & k) == 0)) { a[i0] = hi0; a[i0 | j] = lo0; }
          if (two && (lo1 > hi1) == ((i1 & k) == 0)) { a[i1] = hi1; a[i1 | j] = lo1; }
        }
        wave_sync();
      }
    }
    bool same = false;
    for (int i = lane; i < n; i += 64) {
      const int y = a[i];
      new_col[ob + i] = y;
      same |= i > 0 && a[i - 1] == y;
    }
    if (same) *dup = 1;
    wave_sync();
  }
}

// a listed row of 1025 .. CAP = kRelabelLdsMax entries: the whole workgroup, bitonic network in LDS (a wave per such row: 5.9 ms on the
// symmetric R-MAT-24, the workgroup 3.1).  (A 32768-entry instance in 128 KB of LDS for the hubs was measured too: 14.5 ms of kernel
// at one workgroup per CU and ~190 ms for the first launch of a kernel with more than 64 KB of LDS -- the hubs go through a segmented
// radix sort instead.)
template <int CAP>
__global__ __launch_bounds__(256) void relabel_rows_block_kernel(int nv, const unsigned long long *__restrict__ vsorted, int descending,
                                                                 const int *__restrict__ rp, const int *__restrict__ col,
                                                                 const int *__restrict__ newid, const int *__restrict__ new_rp,
                                                                 int *__restrict__ new_col, int *__restrict__ dup,
                                                                 const int *__restrict__ rows, int cap, const int *__restrict__ long_count) {
  extern __shared__ int lds_row[];
  int *a = lds_row;
  const int tid = threadIdx.x;
  const int count = long_count[1];
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int r = rows[cap - count + q];
    const int v = (int)(unsigned)(vsorted[descending ? nv - 1 - r : r] & 0xffffffffull);
    const int b = rp[v], n = rp[v + 1] - b, ob = new_rp[r];
    if (n > CAP) continue;  // (workgroup-uniform)
    int P = 1024;
    while (P < n) P <<= 1;
    for (int i = tid; i < P; i += 256) a[i] = i < n ? newid[col[b + i]] : 0x7fffffff;
    __syncthreads();
    const int half = P >> 1;
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < half; t += 512) {  // (half >= 512: two exchanges per thread and round)
          const int t2 = t + 256;
          const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1)), i1 = ((t2 & ~(j - 1)) << 1) | (t2 & (j - 1));
          const int lo0 = a[i0], hi0 = a[i0 | j];
          const int lo1 = a[i1], hi1 = a[i1 | j];
          if ((lo0 > hi0) == ((i0 & k) == 0)) { a[i0] = hi0; a[i0 | j] = lo0; }
          if ((lo1 > hi1) == ((i1 & k) == 0)) { a[i1] = hi1; a[i1 | j] = lo1; }
        }
        __syncthreads();
      }
    }
    bool same = false;
    for (int i = tid; i < n; i += 256) {
      const int y = a[i];
      new_col[ob + i] = y;
      same |= i > 0 && a[i - 1] == y;
    }
    if (same) *dup = 1;
    __syncthreads();
  }
}

// rows beyond kRelabelLdsMax entries (the hubs of a symmetric graph: a few thousand rows at most): one workgroup per row writes the
// new ids of its entries, unsorted, into a scratch copy at the row's new place and the row's segment; ONE segmented radix sort over
// those segments alone moves them, sorted, into the copy (PHASE 0).  PHASE 1 looks for two equal neighbours.
template <int PHASE>
__global__ __launch_bounds__(256) void relabel_rows_huge_kernel(int nv, const unsigned long long *__restrict__ vsorted, int descending,
                                                                const int *__restrict__ rp, const int *__restrict__ col,
                                                                const int *__restrict__ newid, const int *__restrict__ new_rp,
                                                                const int *__restrict__ huge_rows, int *__restrict__ scratch,
                                                                int *__restrict__ seg_begin, int *__restrict__ seg_end,
                                                                const int *__restrict__ new_col, int *__restrict__ dup) {
  const int r = huge_rows[blockIdx.x];
  const int v = (int)(unsigned)(vsorted[descending ? nv - 1 - r : r] & 0xffffffffull);
  const int b = rp[v], n = rp[v + 1] - b, ob = new_rp[r];
  if (PHASE == 0) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) scratch[ob + i] = newid[col[b + i]];
    if (threadIdx.x == 0) { seg_begin[blockIdx.x] = ob; seg_end[blockIdx.x] = ob + n; }
  } else {
    bool same = false;
    for (int i = 1 + threadIdx.x; i < n; i += blockDim.x) same |= new_col[ob + i - 1] == new_col[ob + i];
    if (same) *dup = 1;
  }
}

// (dup: set when two sorted keys are equal -- a duplicate entry of a row.  The rows of the copy are ascending by construction, so this
// is all that graph_rows_sorted would look for: the copy is marked without the 0.5 ms pass over its entries.)
__global__ __launch_bounds__(256) void relabel_cols_kernel(long long ne, const unsigned long long *__restrict__ keys, int *__restrict__ col,
                                                           int bits, int *__restrict__ dup) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const unsigned long long k = keys[e];
  col[e] = (int)(unsigned)(k & ((1ull << bits) - 1ull));
  if (e > 0 && keys[e - 1] == k) *dup = 1;
}
// Rows strictly ascending?  Every solver relies on it (bisection, trimmed tasks, position = rank); the reference sorts on request
// (adj_sorted = 0 -> Graph::sort_neighbors, src/common/graph.cc:138).  A descent col[e - 1] >= col[e] is legitimate only where a row
// starts: one pass over the entries counts the descents, one over the rows counts those that sit at the start of a row -- the rows are
// ascending iff the two counts are equal.  (Until round 4 every descent bisected the offsets for its row: one bisection per row of a
// sorted graph, 3.0 ms on R-MAT-24.)
__global__ __launch_bounds__(256) void sorted_check_kernel(long long ne, const int *__restrict__ col, unsigned long long *__restrict__ counts) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long n = 0;
  for (long long e = 1 + (long long)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) n += col[e - 1] >= col[e];
  n = wave_sum_u64(n);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&counts[0], n);
}
__global__ __launch_bounds__(256) void sorted_rowstart_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ col, unsigned long long *__restrict__ counts) {
  unsigned long long n = 0;
  for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < nv; u += (long long)gridDim.x * blockDim.x) {
    const int b = rp[u];
    if (b > 0 && rp[u + 1] > b) n += col[b - 1] >= col[b];
  }
  // (one atomic per wave of a bounded grid: same-address atomics are served one after the other, ~10 ns each -- a wave per 64 rows of
  //  R-MAT-24 spent 2.6 of this kernel's 3.0 ms queueing 260 k of them)
  n = wave_sum_u64(n);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&counts[1], n);
}
// rows are ascending: the numbering is topological (every edge u -> v has u < v) iff no row starts at or below its own vertex
__global__ __launch_bounds__(256) void topo_check_kernel(int nv, const int *__restrict__ rp, const int *__restrict__ col, int *__restrict__ not_topo) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < nv && rp[u + 1] > rp[u] && col[rp[u]] <= u) *not_topo = 1;
}

int graph_rows_sorted(gm_graph *g, bool *out) {
  if (g->sorted_state == 0) {
    if (g->d_rp == nullptr || g->ne < 2) {  // (a handle of >= 2^31 entries is validated when it is oriented)
      g->sorted_state = 1;
    } else {
      HIP_TRY(hipSetDevice(g->device));
      DevBuf<unsigned long long> counts;
      HIP_TRY(counts.alloc(2));
      HIP_TRY(hipMemsetAsync(counts.p, 0, sizeof(unsigned long long) * 2, 0));
      const long long blocks = std::min<long long>((g->ne + 255) / 256, (long long)g->cu_count * 8);
      hipLaunchKernelGGL(sorted_check_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, g->ne, g->d_col, counts.p);
      hipLaunchKernelGGL(sorted_rowstart_kernel, dim3((unsigned)std::min<long long>(((long long)g->nv + 255) / 256, (long long)g->cu_count * 8)), dim3(256), 0, 0, g->nv, g->d_rp, g->d_col, counts.p);
      unsigned long long c[2] = {0, 0};
      HIP_TRY(hipMemcpy(c, counts.p, sizeof(c), hipMemcpyDeviceToHost));
      g->sorted_state = c[0] != c[1] ? 2 : 1;
    }
  }
  *out = g->sorted_state == 1;
  return GM_OK;
}

// "topological" as the task lists use it (an in-edge task streams only the entries BEHIND its own: gm_tables.hip): every edge u -> v has
// u < v AND the rows are ascending (the first entry of a row alone was tested, which says nothing about an unsorted row)
int graph_is_topological(gm_graph *g, bool *out) {
  if (g->topo_state == 0) {
    bool sorted = false;
    const int rc = graph_rows_sorted(g, &sorted);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(g->device));
    DevBuf<int> flag;
    HIP_TRY(flag.alloc(1));
    HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), 0));
    if (g->nv > 0) hipLaunchKernelGGL(topo_check_kernel, dim3((unsigned)((g->nv + 255) / 256)), dim3(256), 0, 0, g->nv, g->d_rp, g->d_col, flag.p);
    int not_topo = 0;
    HIP_TRY(hipMemcpy(&not_topo, flag.p, sizeof(int), hipMemcpyDeviceToHost));
    g->topo_state = (not_topo || !sorted) ? 2 : 1;
  }
  *out = g->topo_state == 1;
  return GM_OK;
}

// mode 0: ids ascending in degree, 1: descending, 2: ascending in (in-degree + out-degree) -- the topological numbering of an oriented
// graph; when the result is not topological after all (a DAG oriented by some other rule) *out is the graph itself.
int get_relabeled(gm_graph *g, int mode, gm_graph **out) {
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->relabel_cache[mode]) { *out = g->relabel_cache[mode]; return GM_OK; }
    if (mode == 2 && g->topo_relabel_failed) { *out = g; return GM_OK; }
  }
  HIP_TRY(hipSetDevice(g->device));
  SetupTimer timer;
  const int nv = g->nv;
  const long long ne = g->ne;
  const size_t nv1 = (size_t)nv + 1, n1 = (size_t)std::max<long long>(ne, 1);
  PoolScope pool(g);  // (the sort buffers: two 64-bit keys per entry + hipCUB's own)
  setup_trace("relabel: begin");
  auto blocks = [](long long n) { return dim3((unsigned)std::max<long long>(1, (n + 255) / 256)); };
  ScanTemp tmp;
  DevBuf<int> indeg, newid, newdeg;
  DevBuf<unsigned long long> vkeys, vsorted, keys, sorted;
  DevBuf<int> dupflag;
  HIP_TRY(dupflag.alloc(1));
  HIP_TRY(hipMemsetAsync(dupflag.p, 0, sizeof(int), 0));
  HIP_TRY(newid.alloc(nv1));
  HIP_TRY(newdeg.alloc(nv1));
  HIP_TRY(vkeys.alloc(nv1));
  HIP_TRY(vsorted.alloc(nv1));
  if (mode == 2 && !g->d_symdeg) {  // (a contended atomic pass: 15 ms on the com-Orkut stand-in; a DAG from gm_graph_orient skips it)
    HIP_TRY(indeg.alloc(nv1));
    HIP_TRY(hipMemsetAsync(indeg.p, 0, sizeof(int) * nv1, 0));
    if (ne > 0) hipLaunchKernelGGL(relabel_indeg_kernel, dim3((unsigned)std::min<long long>((ne + 255) / 256, (long long)g->cu_count * 32)), dim3(256), 0, 0, ne, g->d_col, indeg.p);
  }
  int bits = 1;
  while (bits < 32 && (1ll << bits) < (long long)nv) ++bits;
  if (nv > 0) {
    hipLaunchKernelGGL(relabel_vkeys_kernel, blocks(nv), dim3(256), 0, 0, nv, g->d_rp, (mode == 2 && !g->d_symdeg) ? indeg.p : (const int *)nullptr,
                       mode == 2 ? g->d_symdeg : (const int *)nullptr, vkeys.p);
    size_t bytes = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, vkeys.p, vsorted.p, nv, 0, 64));
    HIP_TRY(tmp.reserve(bytes));
    HIP_TRY(hipcub::DeviceRadixSort::SortKeys(tmp.buf.p, bytes, vkeys.p, vsorted.p, nv, 0, 64));
  }
  hipLaunchKernelGGL(relabel_newid_kernel, blocks((long long)nv1), dim3(256), 0, 0, nv, vsorted.p, mode == 1 ? 1 : 0, g->d_rp, newid.p, newdeg.p);
  setup_trace("relabel: vertex sort");
  gm_graph *r = new gm_graph();
  r->device = g->device;
  r->nv = nv;
  r->ne = ne;
  auto fail = [&](hipError_t e, const char *what) { gm_graph_free(r); return hip_fail(e, what, __FILE__, __LINE__); };
  hipError_t e;
  if ((e = dev_malloc(&r->d_rp, sizeof(int) * nv1)) != hipSuccess) return fail(e, "hipMalloc(rp)");
  if ((e = dev_malloc(&r->d_col, sizeof(int) * n1)) != hipSuccess) return fail(e, "hipMalloc(col)");
  if ((e = dev_exclusive_sum(tmp, newdeg.p, r->d_rp, nv1)) != hipSuccess) return fail(e, "ExclusiveSum");
  // rows of at most kRelabelLdsMax entries are sorted inside the kernels that write them (rank / bitonic network in LDS): no key per
  // entry, no device-wide sort; the few rows beyond that -- the hubs of a symmetric graph -- through one segmented radix sort of their
  // segments (GM_RELABEL_GLOBAL_SORT=1: round 3's 64-bit keys + radix sort of every entry)
  const char *env_gs = gm_opt("GM_RELABEL_GLOBAL_SORT");
  const bool global_sort = env_gs && *env_gs == '1';
  const bool rows_in_lds = !global_sort;
  if (ne > 0 && rows_in_lds) {
    DevBuf<int> long_rows, long_count, huge_rows;
    const int cap = (int)(ne / (kRelabelShort + 1)) + 1, hcap = (int)(ne / (kRelabelLdsMax + 1)) + 1;
    if ((e = long_rows.alloc((size_t)cap)) != hipSuccess || (e = long_count.alloc(3)) != hipSuccess || (e = huge_rows.alloc((size_t)hcap)) != hipSuccess)
      return fail(e, "hipMalloc(long rows)");
    (void)hipMemsetAsync(long_count.p, 0, sizeof(int) * 3, 0);
    const int desc = mode == 1 ? 1 : 0;
    const int wg = std::max(1, std::min((nv + 31) / 32, g->cu_count * 8));
    hipLaunchKernelGGL(relabel_rows_short_kernel, dim3(wg), dim3(256), 0, 0, nv, vsorted.p, desc, g->d_rp, g->d_col, newid.p, r->d_rp, r->d_col, dupflag.p, long_rows.p, cap,
                       long_count.p, huge_rows.p);
    if (g->max_deg > kRelabelShort)
      hipLaunchKernelGGL((relabel_rows_long_kernel<kRelabelMid>), dim3(g->cu_count * 8), dim3(256), 0, 0, nv, vsorted.p, desc, g->d_rp, g->d_col, newid.p, r->d_rp, r->d_col,
                         dupflag.p, long_rows.p, cap, long_count.p);
    if (g->max_deg > kRelabelMid)
      hipLaunchKernelGGL((relabel_rows_block_kernel<kRelabelLdsMax>), dim3(g->cu_count * 8), dim3(256), sizeof(int) * kRelabelLdsMax, 0, nv, vsorted.p, desc, g->d_rp,
                         g->d_col, newid.p, r->d_rp, r->d_col, dupflag.p, long_rows.p, cap, long_count.p);
    if (g->max_deg > kRelabelLdsMax) {  // the hubs of a symmetric graph: their segments through one segmented radix sort
      int nhuge = 0;
      if ((e = hipMemcpy(&nhuge, long_count.p + 2, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
      if (nhuge > 0) {
        DevBuf<int> scratch, seg_begin, seg_end;
        if ((e = scratch.alloc(n1)) != hipSuccess || (e = seg_begin.alloc((size_t)nhuge)) != hipSuccess || (e = seg_end.alloc((size_t)nhuge)) != hipSuccess)
          return fail(e, "hipMalloc(huge rows)");
        hipLaunchKernelGGL((relabel_rows_huge_kernel<0>), dim3((unsigned)nhuge), dim3(256), 0, 0, nv, vsorted.p, desc, g->d_rp, g->d_col, newid.p, r->d_rp, huge_rows.p,
                           scratch.p, seg_begin.p, seg_end.p, (const int *)nullptr, dupflag.p);
        size_t bytes = 0;
        if ((e = hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, bytes, scratch.p, r->d_col, (int)ne, nhuge, seg_begin.p, seg_end.p, 0, bits)) != hipSuccess)
          return fail(e, "SegmentedSortKeys(size)");
        if ((e = tmp.reserve(bytes)) != hipSuccess) return fail(e, "hipMalloc(sort temp)");
        if ((e = hipcub::DeviceSegmentedRadixSort::SortKeys(tmp.buf.p, bytes, scratch.p, r->d_col, (int)ne, nhuge, seg_begin.p, seg_end.p, 0, bits)) != hipSuccess)
          return fail(e, "SegmentedSortKeys");
        hipLaunchKernelGGL((relabel_rows_huge_kernel<1>), dim3((unsigned)nhuge), dim3(256), 0, 0, nv, vsorted.p, desc, g->d_rp, g->d_col, newid.p, r->d_rp, huge_rows.p,
                           (int *)nullptr, (int *)nullptr, (int *)nullptr, (const int *)r->d_col, dupflag.p);
        if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "relabel huge rows");  // (the scratch arrays go out of scope)
      }
    }
    setup_trace("relabel: rows sorted in LDS");
  } else if (ne > 0) {
    if ((e = keys.alloc(n1)) != hipSuccess || (e = sorted.alloc(n1)) != hipSuccess) return fail(e, "hipMalloc(keys)");
    hipLaunchKernelGGL(relabel_keys_kernel, blocks((ne + 3) / 4), dim3(256), 0, 0, nv, ne, g->d_rp, g->d_col, newid.p, bits, keys.p);
  setup_trace("relabel: allocations + entry keys");
    size_t bytes = 0;
    if ((e = hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, keys.p, sorted.p, (int)ne, 0, 2 * bits)) != hipSuccess) return fail(e, "SortKeys(size)");
    if ((e = tmp.reserve(bytes)) != hipSuccess) return fail(e, "hipMalloc(sort temp)");
    if ((e = hipcub::DeviceRadixSort::SortKeys(tmp.buf.p, bytes, keys.p, sorted.p, (int)ne, 0, 2 * bits)) != hipSuccess) return fail(e, "SortKeys");
    hipLaunchKernelGGL(relabel_cols_kernel, blocks(ne), dim3(256), 0, 0, ne, sorted.p, r->d_col, bits, dupflag.p);
  }
  setup_trace("relabel: entry sort + columns");
  if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "relabel kernels");
  {
    int dup = 0;
    if ((e = hipMemcpy(&dup, dupflag.p, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return fail(e, "hipMemcpy");
    r->sorted_state = dup ? 2 : 1;
  }
  int rc = finish_handle(r);
  if (rc) { gm_graph_free(r); return rc; }
  r->max_deg = g->max_deg;  // (a permutation of the same rows)
  r->pool_owner = g;
  setup_trace("relabel: finish_handle");
  if (mode == 2) {
    bool topo = false;
    rc = graph_is_topological(r, &topo);
    if (rc) { gm_graph_free(r); return rc; }
    if (!topo) {  // not an orientation by (degree, id): nothing gained, keep the graph as given
      gm_graph_free(r);
      std::lock_guard<std::mutex> lk(g->mu);
      g->topo_relabel_failed = true;
      g->setup.relabel_ms += timer.ms();
      *out = g;
      return GM_OK;
    }
  }
  std::lock_guard<std::mutex> lk(g->mu);
  g->relabel_cache[mode] = r;
  g->setup.relabel_ms += timer.ms();
  *out = r;
  return GM_OK;
}

