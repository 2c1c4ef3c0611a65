// gm_launch.hip -- the solvers of the C ABI: launch prologue / epilogue, run_pattern (TC, diamond, 3-motif, k-clique and the per-edge sums
// of 4-motif: tables, shares, class launches), the SgL map / flat / nested launches, gm_tc / gm_sgl / gm_clique / gm_motif*.
// Reference launch logic: src/triangle/gpu_base.cu:36-45, src/sgl/gpu_base.cu:37-75, src/clique/gpu_base.cu:28-50, src/motif/gpu_base.cu:42-75.
#include "gm_host.h"

using namespace gm;

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

// a big handle (ne >= 2^31, 64-bit offsets): the mining kernels index the graph they walk with 32 bits
static int reject_big(const gm_graph *g) {
  if (!g || !g->d_rp64) return GM_OK;
  g_last_error = "this solver walks the graph it is given with a 32-bit task index; " + std::to_string(g->ne) +
                 " entries: orient it (gm_graph_orient) for TC / k-clique, gm_motif k = 3 counts through the formula solver";
  return GM_ERR_TOO_LARGE;
}

// sum of the lengths of the rows beyond kStageCapBig entries (the task edges of the giant-row kernel), once per graph
__global__ __launch_bounds__(256) void giant_edges_kernel(int nv, const int *__restrict__ rp, unsigned long long *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
    const int d = rp[v + 1] - rp[v];
    if (d > kStageCapBig) s += (unsigned long long)d;
  }
  s = gm::wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}
static int giant_task_edges(gm_graph *g, unsigned long long *out) {
  if (g->giant_edges == ~0ull) {
    HIP_TRY(hipSetDevice(g->device));
    unsigned long long *d_s = nullptr, s = 0;
    HIP_TRY(dev_malloc(&d_s, 8));
    hipError_t e = hipMemset(d_s, 0, 8);
    if (e == hipSuccess && g->nv > 0)
      hipLaunchKernelGGL(giant_edges_kernel, dim3((unsigned)std::min<long long>(((long long)g->nv + 255) / 256, 2048)), dim3(256), 0, 0, g->nv, g->d_rp, d_s);
    if (e == hipSuccess) e = hipMemcpy(&s, d_s, 8, hipMemcpyDeviceToHost);
    dev_free(d_s);
    if (e != hipSuccess) return hip_fail(e, "giant_edges_kernel", __FILE__, __LINE__);
    g->giant_edges = s;
  }
  *out = g->giant_edges;
  return GM_OK;
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
  if (int rc = reject_big(cg)) return rc;
  c.g = const_cast<gm_graph *>(cg);
  memset(&c.la, 0, sizeof c.la);
  if (la) c.la = *la;
  c.world = c.la.world > 1 ? c.la.world : 1;
  c.rank = c.la.rank;
  if (c.rank < 0 || c.rank >= c.world) return GM_ERR_INVALID;
  if (!h_out && !c.la.d_counts) return GM_ERR_INVALID;
  HIP_TRY(hipSetDevice(c.g->device));
  {  // every solver relies on ascending rows (checked once per handle; the reference sorts on request: adj_sorted = 0)
    bool sorted = false;
    if (int rc = graph_rows_sorted(c.g, &sorted)) return rc;
    if (!sorted) {
      g_last_error = "the neighbour lists of this graph are not strictly ascending: call gm_graph_sort_neighbors first (Graph::sort_neighbors, adj_sorted = 0)";
      return GM_ERR_INVALID;
    }
  }
  c.stream = (hipStream_t)c.la.stream;
  HIP_TRY(hipMemsetAsync(c.g->d_counters, 0, 64, c.stream));
  return GM_OK;
}

static int start_timer(LaunchCtx &c) {
  c.g->ring_alias = nullptr;
  c.g->ring_extra[0] = c.g->ring_extra[1] = nullptr;
  c.evp = c.g->ev[c.g->ev_launches % gm_graph::kEvRing];
  c.g->ev_corner[c.g->ev_launches % gm_graph::kEvRing] = false;
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

int run_pattern(Pattern pat, const gm_graph *cg, const gm_launch *la, int k, uint64_t *h_out, int nout, gm_stats *st, int fin_mode,
                unsigned long long fin_base, unsigned *sup_out) {
  // edge supports + sum C(t, 2) (gm_sup.hip): the triangle pass of the task lists with another match handler -- everything up to the
  // launch is the triangle count's
  // (PAT_SUPPORT_PART: the share of a rank -- any world -- of the supports only, added into the caller's zeroed buffer: the ranks' arrays
  // are summed by a reduce-scatter and gm_diamond_support_finish takes sum C(t, 2) of a slice)
  const bool sup_part = pat == PAT_SUPPORT_PART;
  const bool support = pat == PAT_SUPPORT || sup_part;
  if (sup_part && !sup_out) return GM_ERR_INVALID;
  if (support) pat = PAT_TC;
  if (fin_mode < 0) fin_mode = (pat == PAT_MOTIF3) ? FIN_MOTIF3 : FIN_COPY;
  LaunchCtx ctx;
  int rc0 = begin_launch(cg, la, h_out, ctx);
  if (rc0) return rc0;
  gm_graph *g = ctx.g;
  la = &ctx.la;
  const int world = ctx.world, rank = ctx.rank;
  hipStream_t stream = ctx.stream;
  setup_trace("run_pattern: begin_launch");

  // tune[0] = chunk target override, tune[1] = grab, tune[2] = cost_x_step, tune[3] = cost_y_step,
  // tune[4] = blocks per CU override, tune[5] = force "search in HBM" (no LDS staging) when 1
  // default chunk size: as large as the LDS stage allows (fewer dequeues, better staging reuse) while every rank still
  // gets >= ~2 chunks per resident workgroup for the dynamic dequeue to balance (matters for strong scaling at N = 8)
  // (round 3: 4 chunks per resident workgroup for a rank of a larger job -- with ~2 the heaviest-first dequeue of a 1/8 share of the
  // LiveJournal stand-in ended 50 % above its mean, TC 1.01 ms per rank against 0.66 ideal; one rank keeps the round-2 rule)
  int target = kDefaultChunk;
  long long min_chunks = (world > 1 ? 4LL : 2LL) * g->cu_count * 7;
  if (const char *e = gm_sweep_env("GM_MIN_CHUNKS_PER_CU")) min_chunks = (long long)std::max(1, atoi(e)) * g->cu_count;  // (sweeps)
  while (target > 128 && g->ne / ((long long)world * target) < min_chunks) target >>= 1;
  if (la->chunk > 0) target = la->chunk;
  if (la->tune[0] > 0) target = la->tune[0];
  target = std::max(64, std::min(target, kStageCap));
  const bool clique = pat == PAT_CLIQUE4 || pat == PAT_CLIQUEK;
  ChunkTable *tab = nullptr;
  // tune[6] & 0x1000 (tests): cut every chunk above 4096 estimated entries into parts
  // chunk costs are estimated keys (DAG patterns: d(u) + d(v) per edge; symmetric patterns: streamed keys, bitmap probes
  // weighted kProbeCost); parts bound the longest task of a launch
  // TC: the shorter list of every edge is streamed against the longer one (gm_tch.hip) when every DAG row fits the LDS stage
  // (tune[6] & 0x4000000: A/B switch, the chunked kernel that streams N+(v) of every out-edge).  Its chunks host the tasks of
  // their vertices -- a hub hosts 10^5 in-edges -- so their cost is counted from the task lists and heavy chunks are cut into
  // parts of 1 M keys / world (>= 128 K): one-GPU simulation of an 8-rank share of R-MAT-22, parts of 8 M / 512 K / 128 K / 32 K keys:
  // 5.36 / 1.18 / 0.99 / 1.15 ms per rank (one GPU: 6.50 / 6.54 / 6.73 / 8.09 ms), profiles/r02/ab_tct_part_cap.log
  const bool use_tct = pat == PAT_TC && !(la->tune[6] & 0x4000000) && la->tune[5] != 1 && g->ne > 0 && !gm_sweep_env("GM_HOST_TABLES");
  // The 2048-entry stage costs occupancy (four instead of six workgroups per CU, R-MAT-22 on it: 3.04 vs 2.57 ms), and only the few hosts with
  // rows of 1025 .. 2048 entries need it: a graph that has such rows runs TWO tables -- hosts with rows <= 1024 on the 1024-entry kernel,
  // the others on the 2048-entry one (own dequeue word).  GM_TCT_STAGE_BIG: the 2048-entry stage for every host (A/B).
  const bool stage_big_all = gm_sweep_env("GM_TCT_STAGE_BIG") != nullptr;
  bool split_stage = use_tct && g->max_deg > kStageCap && !stage_big_all && !gm_sweep_env("GM_TCT_NO_SPLIT_STAGE");
  int tct_stage = ((g->max_deg <= kStageCap && !stage_big_all) || split_stage) ? kStageCap : kTctStageMax;
  // (rows beyond the 2048-entry stage host nothing: their out-edges are the tasks of the chunked kernel, on a table of those rows only)
  const bool tct_long = use_tct && g->max_deg > kTctStageMax;
  const unsigned long long tct_part = use_tct ? task_part_cap(g, world) : 0ull;
  const unsigned long long part_cap = (la->tune[6] & 0x1000) ? 4096ull : (stage_cap_of(pat) != kStageCapWide
       ? (use_tct ? tct_part : kPartCostCap)
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
  if (support && (!use_tct || (world > 1 && !sup_part))) return GM_ERR_UNSUPPORTED;  // (the caller takes the per-edge kernels)
  if (support && tct_long) {  // the out-edges of the rows beyond the stage: sup_long_kernel, below
    const int rc_l = ensure_long_rows(g);
    if (rc_l) return rc_l;
  }
  // the triangle count reads the key stream + the lists of the longer tasks (ensure_keystream); the edge supports, the kernels without the
  // stream (tune[6] & 0x20000000) and the handles that cannot have one read the full task lists
  bool use_kst = false;
  if (use_tct) {
    // The edge supports take the stream only where matches are RARE: a match of a streamed key costs them two gathers (the entries beside
    // the stream) and two global atomics, where a task of the lists costs one atomic per match and one per task.  Flat / power-law LJ-size
    // 1.06 / 1.55 -> 0.70 / 1.32 ms, but R-MAT-22 8.50 -> 10.95 and R-MAT-24 102 -> 115: the switch is the one that decides the renumbering
    // (sum d+^2 / |E+| below kSupStreamMaxMeanRow: short lists) -- and few triangles per edge, which short lists do not imply: on a graph
    // with LiveJournal's size AND triangle density (planted communities, 6.7 triangles per DAG entry; round 6) the stream's two atomics per
    // match made the diamond 13.4 ms where the task lists take 2.8.  The density is estimated from a sample of the entries
    // (ensure_tri_per_edge): the stream below 0.5 triangles per entry (power law: 0.1, flat: 0).  GM_SUP_STREAM=0 / 1 forces it.
    bool sup_stream = false;
    if (support && !(la->tune[6] & 0x20000000)) {
      if (const char *e = gm_opt("GM_SUP_STREAM")) sup_stream = atoi(e) != 0;
      else sup_stream = ensure_mean_sq_deg(g) == GM_OK && g->mean_sq_deg < (double)kSupStreamMaxMeanRow && ensure_tri_per_edge(g) == GM_OK &&
                        g->tri_per_edge < 0.5;
    }
    if ((sup_stream || !support) && !(la->tune[6] & 0x20000000)) {
      const int rc_k = ensure_keystream(g, support, &use_kst, !support && !tct_long);  // (the triangle count: the hub corner may stay out, gm_ctc.hip)
      if (rc_k) return rc_k;
    }
    if (!use_kst) {
      const int rc_t = ensure_tasklists(g, support);
      if (rc_t) return rc_t;
    }
  }
  if (support && !sup_part && !g->d_sup) HIP_TRY(dev_malloc(&g->d_sup, sizeof(unsigned) * (size_t)std::max<long long>(g->ne, 1)));
  // match masks instead of one atomic per streamed edge (gm_sup.hip): one GPU, the task lists, a topologically numbered DAG
  // (tune[6] & 0x40000000: A/B switch, every streamed edge by an atomic)
  bool sup_masks = false;
  if (support && !sup_part && world == 1 && use_tct && !use_kst && !(la->tune[6] & 0x40000000)) {
    const int rc_m = ensure_sup_masks(g);
    if (rc_m) return rc_m;
    sup_masks = g->smask_state == 1;
  }
  setup_trace("run_pattern: task lists");
  // the hosts with rows of 1025 .. 2048 entries, on the 2048-entry kernel -- when a rank's share of them can fill the chip about twice: the two
  // launches follow each other on the stream, and the second waits for the first one's last chunk.  R-MAT-24 has 8.5 K such chunks with a
  // third of all keys: split / one table, ms per rank at world 1 / 2 / 4 / 8: TC 32.3 / 16.1 / 9.26 / 5.20 against 36.2 / 18.2 / 9.30 / 4.82,
  // edge supports - / 51.0 / 26.0 / 16.7 against - / 60.5 / 30.4 / 15.5 (profiles/r04/ab_split_stage.txt).  GM_TCT_SPLIT_ALWAYS: at every world.
  ChunkTable *tab_big = nullptr;
  if (split_stage) {
    RowFilter rb;
    rb.tct = (use_kst && !support && g->kst_skip_from < g->nv) ? 2 : 1;  // (2: priced from the stream without the hub corner, gm_tables.hip)
    rb.only_lo = kStageCap;
    rb.only_hi = kTctStageMax;
    const int rc_b = get_table(g, target, true, 0, part_cap, kTctStageMax, &tab_big, rb, kBitmapMinDeg);
    if (rc_b) return rc_b;
    if ((long long)tab_big->n / world < 2ll * g->cu_count * 4 && !gm_opt("GM_TCT_SPLIT_ALWAYS")) {
      tab_big = nullptr;
      split_stage = false;
      tct_stage = kTctStageMax;
    }
  }
  // the triangles of the hub corner on the matrix cores (gm_ctc.hip): this handle's key stream holds no task of the corner's rows
  const int corner_from = use_kst ? g->kst_skip_from : (use_tct ? g->tl_skip_from : 0x7fffffff);  // the rows the stream / the task lists in use leave out
  const bool tc_core = use_tct && !support && corner_from < g->nv;
  const bool sup_core = support && !use_kst && corner_from < g->nv;
  if (sup_core) {
    const int rc_c = ensure_sup_corner(g);
    if (rc_c) return rc_c;
  }
  RowFilter rf;
  rf.tct = use_tct ? ((tc_core && use_kst) ? 2 : 1) : 0;
  if (tct_long) { rf.skip_lo = kTctStageMax; rf.skip_hi = 0x7fffffff; }
  if (split_stage) { rf.skip_lo = kStageCap; rf.skip_hi = 0x7fffffff; }  // (this table: the hosts whose rows fit the 1024-entry stage)
  // 4-clique: the first level is re-hosted (gm_cbuild.hip) for every vertex whose row fits its stage -- the narrow chunk table and
  // the wide list live in the CliquePlan; what is left for THIS table are the rows beyond kCbMaxDeg (mine_kernel's arena path)
  if (use_wide) rf.only_lo = kCbMaxDeg;
  // Rows of 1025..3072 entries fit the general kernel's stage, but their partner lists (mean 600 keys on R-MAT-24) are cheaper
  // against a hashed set than against the filter + bisection of a multi-row chunk: measured (profiles/r02/ab_hrow_class_lower_bound.log,
  // ms, lower bound 3072 / 2048 / 1024 / 512 / 256) diamond R-MAT-24 262 / 240 / 239 / 238 / 237, R-MAT-22 21.1 / 21.0 / 17.5 / 17.6 / 17.5,
  // 3-motif R-MAT-24 185 / 172 / 172 / 171 / 171.
  int cls_lo = kClassRowMin;
  if (const char *e = gm_sweep_env("GM_CLS_LO")) cls_lo = std::max(64, atoi(e));  // (sweeps)
  // giant rows (> kStageCapBig entries): hashed sets of row pieces (giant_kernel, gm_hrow.hip) instead of SPLIT chunks probing
  // dense bitmaps in HBM (tune[6] & 0x1000000: A/B switch, they stay SPLIT chunks of the general kernel)
  const bool use_range = use_classes && !(la->tune[6] & 0x1000000) && hrow_fits(g->nv, 2);  // (its pieces are class-2 sets: nv <= 2^27)
  if (use_classes) { rf.skip_lo = cls_lo; rf.skip_hi = use_range ? 0x7fffffff : kStageCapBig; }
  int rc = get_table(g, target, !clique, clique ? kBitWords : 0, part_cap, use_tct ? tct_stage : stage_cap_of(pat), &tab, rf, use_classes ? kStageCapBig : kBitmapMinDeg);
  if (rc) return rc;
  ChunkTable *tab_long = nullptr;
  if (tct_long && !support) {
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
    if (const char *e = gm_sweep_env("GM_CLS_CAP_MKEYS")) cls_cap = (unsigned long long)std::max(1, atoi(e)) << 20;  // (sweeps)
    if (const char *e = gm_sweep_env("GM_CLS_CAP_KKEYS")) cls_cap = (unsigned long long)std::max(16, atoi(e)) << 10;
    // (target 1: every row is a chunk of its own -- the class kernels take one-row chunks -- also below the general kernel's chunk target)
    rc = get_table(g, 1, false, 0, cls_cap, kStageCapMid, &tab_cls[1], r1, 0x7fffffff);
    if (rc) return rc;
    rc = get_table(g, 1, false, 0, cls_cap, kStageCapBig, &tab_cls[2], r2, 0x7fffffff);
    if (rc) return rc;
    if (use_range) {  // pieces of <= kGiantEdges task edges, never cut into parts (part_cap 0: giant_kernel ignores part / nparts)
      RowFilter r3;
      r3.only_lo = kStageCapBig;
      // A chunk is the unit of the dequeue and costs its whole row's set builds (~20 us per piece of 24576 entries): as large as
      // kGiantEdges where there are many, smaller where the giant rows of the graph (or of a rank's share) would otherwise be a
      // handful of chunks -- the LiveJournal stand-in has ~40 chunks of 8192 edges, 3.9 ms each: the whole launch of a 1/8 share
      // waited for ONE of them (1.9 ms ideal). Aim at 4 chunks per resident workgroup and rank.
      int gtarget = kGiantEdges;
      {
        unsigned long long ge = 0;
        rc = giant_task_edges(g, &ge);
        if (rc) return rc;
        const unsigned long long per = ge / ((unsigned long long)g->cu_count * (unsigned long long)giant_per_cu() * 4ull * (unsigned long long)world);
        gtarget = (int)std::max<unsigned long long>(512, std::min<unsigned long long>((unsigned long long)kGiantEdges, per));
        gtarget = (gtarget + 63) & ~63;
        if (const char *e = gm_sweep_env("GM_GIANT_TARGET")) gtarget = std::max(64, std::min(atoi(e), kGiantEdges));  // (sweeps)
      }
      rc = get_table(g, gtarget, true, 0, 0ull, kStageCapBig, &tab_cls[3], r3, 0x7fffffff);
      if (rc) return rc;
    }
  }
  CliquePlan *plan = nullptr;
  if (use_wide) {
    rc = get_clique_plan(g, rank, world, la->policy, target, part_cap, &plan);
    if (rc) return rc;
  }

  setup_trace("run_pattern: tables / plan");
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
    p.g.tedge = support ? g->d_tedge : nullptr;
    if (sup_masks) {
      p.g.tmoff = g->d_tmoff;
      p.smask = g->d_smask;
    }
    // the triangle count streams the short lists from their task-major copies (gm_host.h d_colk; tune[6] & 0x20000000: from their rows);
    // the edge supports need the entries of the streamed keys in col[] itself
    if (use_kst && support) {  // ... with the entries the supports need beside them (the second set when the stream was built without)
      p.g.kst = g->d_kst2 ? g->d_kst2 : g->d_kst;
      p.g.kst_rp = g->d_kst_rp;
      p.g.kst_et = g->d_kst_et;
      p.g.trp = g->d_trpl;
      p.g.tdesc = g->d_tdescl2 ? g->d_tdescl2 : g->d_tdescl;
      p.g.tedge = g->d_tedgel;
    } else if (use_kst) {  // short lists as one tagged key stream, the longer ones as tasks
      p.g.kst = g->d_kst;
      p.g.kst_rp = g->d_kst_rp;
      p.g.trp = g->d_trpl;
      p.g.tdesc = g->d_tdescl;
    } else if (!support && g->d_colk && g->d_tdesck && !(la->tune[6] & 0x20000000)) {
      p.g.col = g->d_colk;
      p.g.tdesc = g->d_tdesck;
    }
  }
  unsigned long long my_edges = 0;
  // this rank's share of a table: chunk ids first + i*step of the dequeue order (or a contiguous / vertex range)
  int share_rc = GM_OK;  // (take_share: a failed fetch of a table's host views)
  auto take_share = [&](ChunkTable *tb, MineParams &q) {
    q.chunks = tb->d;
    q.chunk_slot = tb->d_slot;
    q.bitmaps = tb->d_bitmaps;
    q.bitmap_words = tb->bitmap_words;
    q.row_slot = tb->d_row_slot;
    const long long n = (long long)tb->n;
    long long first = 0, step = 1, count = 0;
    // dequeue order (tune[6] & 0x4000: plain chunk-id order; & 0x2000: swap the two orders -- ablation only)
    // measured on R-MAT (one rank): the cliques want the full cost order (4-clique 220.6 -> 208.6 ms, 5-clique 796 -> 589 ms),
    // the symmetric-graph patterns the locality-preserving heavy-first order at every world size, TC heavy-first for one
    // rank and the full order for shares
    const bool clique_pat = pat == PAT_CLIQUE4 || pat == PAT_CLIQUEK;
    const int which = (((world > 1 && !sym_pat) || clique_pat) ? 1 : 0) ^ ((la->tune[6] & 0x2000) ? 1 : 0);
    const bool lpt = la->policy == GM_PART_ROUND_ROBIN && tb->d_order[which] && !(la->tune[6] & 0x4000);
    q.order = lpt ? tb->d_order[which] : nullptr;
    if (world > 1 && la->policy != GM_PART_VERTEX && tb->d_edges) {
      // a rank of a larger job: its records as a list of their own -- every part of a chunk with the same rank (ShareOrder, gm_host.h)
      const ShareOrder *so = nullptr;
      if (int rcs = get_share_order(g, tb, world, rank, la->policy, lpt ? which : -1, &so)) { share_rc = rcs; return; }
      q.first = 0;
      q.step = 1;
      q.count = (int)so->n;
      q.order = so->d;
      my_edges += so->edges;
      return;
    }
    if (la->policy == GM_PART_VERTEX) {  // contiguous chunk range whose first vertex lies in this rank's vertex range
      if (int rcv = table_host_views(g, tb)) { share_rc = rcv; return; }
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
    // task edges of the share (gm_stats): the whole table's total came with the table; a share reads the per-chunk host views
    if (first == 0 && step == 1 && count == n) { my_edges += tb->total_edges; return; }
    if (int rcv = table_host_views(g, tb)) { share_rc = rcv; return; }
    if (step == 1) my_edges += tb->edge_prefix[first + count] - tb->edge_prefix[first];  // (any order: the same set)
    else for (long long j = first; j < n; j += step) {
      const size_t c = lpt ? (size_t)tb->order[which][(size_t)j] : (size_t)j;
      my_edges += tb->edge_prefix[c + 1] - tb->edge_prefix[c];
    }
  };
  take_share(tab, p);
  if (share_rc) return share_rc;
  p.grab = la->tune[1] > 0 ? la->tune[1] : 1;
  // the task-list kernels take TWO chunks per dequeue where a resident workgroup has many to take (one device atomic + one workgroup
  // barrier less per chunk: flat LJ-size TC 0.605 -> 0.427 ms, power law 0.77 -> 0.71, R-MAT-22 2.57 -> 2.48; four: 0.423 / 0.71 / 3.37 --
  // the heavy chunks at the head of the queue then pair up, profiles/r04/ab_tc_grab.txt)
  // (not the edge supports: their launch time is the atomics', and a rank's share loses -- diamond R-MAT-22 at 2 / 4 ranks 6.06 / 4.79 ms with two
  // chunks per dequeue against 4.44 / 3.13 with one)
  if (la->tune[1] <= 0 && use_tct && !support && p.count >= 8ll * (long long)g->cu_count * (long long)(tct_stage <= kStageCap ? 6 : 4)) p.grab = 2;
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
    const unsigned long long region = tab->max_bit_words + 4096ull;
    const int plist = (g->max_deg + 64) & ~63;  // (rows beyond 4096 columns keep their position lists here: cliquek_count_sub_any)
    const unsigned long long slot_words = (pat == PAT_CLIQUEK) ? (unsigned long long)(k - 2) * (region + (unsigned long long)plist) : tab->max_bit_words;
    p.scratch_region = region;
    p.scratch_plist = plist;
    {  // a slot per workgroup: with very long rows fewer workgroups, so that the arena stays within half of the free memory
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const unsigned long long budget = (unsigned long long)(free_b + g->scratch_bytes) / 2ull;
      const unsigned long long fit = budget / std::max<unsigned long long>(slot_words * sizeof(unsigned), 1ull);
      if (fit == 0) {
        g_last_error = "k-clique: the bit-matrix of the longest DAG row (" + std::to_string(g->max_deg) + " entries) does not fit the device memory";
        return GM_ERR_TOO_LARGE;
      }
      if ((unsigned long long)grid > fit) grid = (int)fit;
    }
    const size_t need = (size_t)slot_words * sizeof(unsigned) * (size_t)grid;
    if (need > g->scratch_bytes) {
      if (g->d_scratch) dev_free(g->d_scratch);
      g->d_scratch = nullptr;
      g->scratch_bytes = 0;
      HIP_TRY(dev_malloc(&g->d_scratch, need));
      g->scratch_bytes = need;
    }
    p.scratch = g->d_scratch;
    p.scratch_words = slot_words;
  }

#ifdef GM_DEBUG_CHUNKS
  unsigned long long *d_ticks = nullptr;
  HIP_TRY(dev_malloc(&d_ticks, sizeof(unsigned long long) * std::max<size_t>(tab->n, 1)));
  HIP_TRY(hipMemset(d_ticks, 0, sizeof(unsigned long long) * std::max<size_t>(tab->n, 1)));
  p.chunk_ticks = d_ticks;
#endif
  // (a side stream is a hardware queue: ~17 ms each to create -- GM_SETUP_TRACE, the first 4-clique call spent 35 ms here for a variant that is off)
  if (use_wide && plan && plan->core_base >= 0 && gm_sweep_env("GM_CLIQUE_SIDE_STREAM")) {  // (4-clique A/B: the gathered build beside the streamed one, below)
    for (int i = 0; i < 2; ++i) {
      if (!g->aux_stream[i]) {
        HIP_TRY(hipStreamCreateWithFlags(&g->aux_stream[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g->aux_done[i], hipEventDisableTiming));
      }
    }
  }
  if (use_classes) {
    // what the class launches may allocate, before the timer starts: the giant-row kernel's scratch -- per workgroup, where the pieces of
    // the row cut the partner lists of its chunk, kGiantEdges ints per piece (giant_bounds).  (The side streams are created by the first
    // launch that sends a class to one -- a class whose chunks cannot fill the chip: small graphs, shares of a rank -- and that launch's
    // time then includes it; creating all three up front cost every first call ~50 ms.)
    if (tab_cls[3] && tab_cls[3]->n > 0) {
      const size_t need = (size_t)giant_scratch_words(g->max_deg) * sizeof(unsigned) * (size_t)g->cu_count * (size_t)giant_per_cu();
      if (need > g->scratch_bytes) {
        if (g->d_scratch) dev_free(g->d_scratch);
        g->d_scratch = nullptr;
        g->scratch_bytes = 0;
        HIP_TRY(dev_malloc(&g->d_scratch, need));
        g->scratch_bytes = need;
      }
    }
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  uint64_t plan_chunks = 0;
  if (use_wide && plan) {
    // 4-clique, re-hosted: per arena round, (1) cbuild_kernel over the round's host chunks builds every row of every owner's matrix,
    // (2) clique_small_kernel counts the matrices of the narrow chunks, the big-LDS classes X / L / S those of the wide vertices
    {  // task edges of the share: the entries of its narrow chunks + the rows of its wide vertices
      ChunkTable *tn = plan->tabN;
      if (plan->n_count == (long long)tn->n) {
        my_edges += tn->total_edges;
      } else {
        rc = table_host_views(g, tn);
        if (rc) return rc;
      }
      for (long long i = 0; i < plan->n_count && plan->n_count != (long long)tn->n; ++i) {
        const long long pos = plan->n_first + i * plan->n_step;
        const size_t c = plan->d_order ? (size_t)tn->order[plan->order_which][(size_t)pos] : (size_t)pos;
        my_edges += tn->edge_prefix[c + 1] - tn->edge_prefix[c];
      }
      my_edges += plan->wide_edges;
    }
    HIP_TRY(hipMemsetAsync(g->d_wide_queue, 0, 65536, stream));
    setup_trace("launch: clique prologue");
    const bool prof = gm_sweep_env("GM_WIDE_PROFILE") != nullptr;
    int qword = 0;
    for (const auto &rd : plan->rounds) {
      if (qword + 9 > 16384) return GM_ERR_TOO_LARGE;  // (more than ~3000 arena rounds)
      bool gather_joined = false;
      // the rows of the wide vertices whose first endpoint lies in the hub core: gathered (gm_cgather.hip) -- by blocks of core rows resident in
      // LDS where the plan lists the round's (vertex, block) units (round 6), row by row from the bitmap otherwise (tune[6] & 0x8000000: on request)
      const bool blocked = plan->core_base >= 0 && rd.w1 > rd.w0 && rd.d_gunits != nullptr && rd.n_gitems > 0 && !(la->tune[6] & 0x8000000);
      if (blocked) {
        CGatherBParams cb;
        memset(&cb, 0, sizeof cb);
        cb.mat = g->d_wide_mat;
        cb.tri = g->d_cg_tri;
        cb.rowbase = g->d_cg_rowbase;
        cb.blk = g->d_cg_blk;
        cb.tab = rd.d_gtab;
        cb.units = rd.d_gunits;
        cb.items = rd.d_gitems;
        cb.delta = g->core_base & 31;
        cb.count = (int)rd.n_gitems;
        cb.queue = g->d_wide_queue + qword++;
        const int ggrid = (int)std::max<long long>(1, std::min<long long>(cb.count, (long long)g->cu_count * cgatherb_per_cu()));
        HIP_TRY(launch_cgatherb(cb, ggrid, stream));
        setup_trace("launch: core gather (blocked)");
      } else if (plan->core_base >= 0 && rd.w1 > rd.w0) {
        CGatherParams cg;
        memset(&cg, 0, sizeof cg);
        cg.rp = g->d_rp;
        cg.col = g->d_col;
        cg.verts = plan->d_verts;
        cg.base = plan->d_slot_base;
        cg.mat = g->d_wide_mat;
        cg.core = g->d_core;
        cg.core_base = g->core_base;
        cg.core_words = (g->core_h + 31) / 32;
        cg.core_bytes = (unsigned long long)g->core_h * (unsigned long long)cg.core_words * 4ull;
        cg.first_slot = (int)rd.w0;
        cg.count = (int)(rd.w1 - rd.w0);
        cg.queue = g->d_wide_queue + qword++;
        const int ggrid = (int)std::max<long long>(1, std::min<long long>(cg.count, (long long)g->cu_count * cgather_per_cu()));
        // (GM_CLIQUE_SIDE_STREAM=1: on a side stream beside the streamed build and the counts of the narrow vertices -- they touch other rows
        // of the arena.  Measured and not taken: the gathers want all 32 waves of a CU -- 4 / 3 / 2 / 1 workgroups per CU beside the streamed
        // build: 29.2 / 29.5 / 33.8 / 53.4 ms against 27.7 one after the other, profiles/r04/ab_clique4_side_stream.txt)
        const bool side = gm_sweep_env("GM_CLIQUE_SIDE_STREAM") != nullptr;
        hipStream_t gs = side ? g->aux_stream[0] : stream;
        if (side) {
          HIP_TRY(hipEventRecord(g->aux_done[1], stream));  // after the queue words were zeroed / the previous round's counts read the arena
          HIP_TRY(hipStreamWaitEvent(gs, g->aux_done[1], 0));
        }
        HIP_TRY(launch_cgather(cg, ggrid, gs));
        setup_trace("launch: core gather");
        if (side) {
          HIP_TRY(hipEventRecord(g->aux_done[0], gs));
          gather_joined = true;
        }
      }
      if (rd.n_tasks > 0) {
        CBuildParams pw;
        memset(&pw, 0, sizeof pw);
        pw.g = p.g;
        pw.chunks = rd.host_tab.d;
        pw.order = rd.host_tab.d_order[1];  // heaviest host chunks first
        // (a chunk of hosts without a task of this rank's owners has cost 0 and sits at the end of the cost order: not dequeued at all --
        // a rank of eight found its 7 M tasks in 97.5 K chunks, most of them empty for it)
        pw.count = (int)std::min(rd.host_tab.n, rd.host_tab.n_with_cost);
        pw.trp = rd.d_trp;
        pw.tasks = rd.d_tasks;
        pw.queue = g->d_wide_queue + qword++;
        pw.mat = g->d_wide_mat;
        pw.flags = p.flags;
        const int bgrid = (int)std::max<long long>(1, std::min<long long>(pw.count, (long long)g->cu_count * cbuild_per_cu(plan->stage)));
        if (pw.count > 0) HIP_TRY(launch_cbuild(pw, plan->stage, bgrid, stream));
        setup_trace("launch: streamed build");
        plan_chunks += (uint64_t)pw.count;
      }
      if (rd.n_count > 0) {
        CliqueSmallParams cs;
        memset(&cs, 0, sizeof cs);
        cs.rp = g->d_rp;
        cs.chunks = plan->tabN->d;
        cs.order = plan->d_order;
        cs.first = (int)(plan->n_first + rd.n_pos0 * plan->n_step);
        cs.step = (int)plan->n_step;
        cs.count = (int)rd.n_count;
        cs.base = rd.d_base;
        cs.mat = g->d_wide_mat;
        cs.queue = g->d_wide_queue + qword++;
        cs.counters = g->d_counters;
        cs.topo = plan->topo ? 1 : 0;
        const int sgrid = (int)std::max<long long>(1, std::min<long long>(cs.count, (long long)g->cu_count * 8));
        HIP_TRY(launch_clique_small(cs, sgrid, stream));
        setup_trace("launch: narrow counts");
        plan_chunks += (uint64_t)rd.n_count;
      }
      if (gather_joined) HIP_TRY(hipStreamWaitEvent(stream, g->aux_done[0], 0));  // the wide vertices' rows are complete
      // pair counts of the wide vertices on the matrix cores (gm_cmma.hip)
      for (int cls = 2; cls >= 0; --cls) {  // the column blocks and the one-per-CU workgroups before the small ones
        CliqueCountParams c;
        memset(&c, 0, sizeof c);
        c.rp = g->d_rp;
        c.verts = plan->d_verts;
        c.base = plan->d_slot_base;
        c.mat = g->d_wide_mat;
        c.slots = plan->d_mcls_slots + rd.mcls_begin[cls];
        c.qrec = plan->d_mcls_rec + rd.mcls_begin[cls];
        c.count = (int)(rd.mcls_begin[cls + 1] - rd.mcls_begin[cls]);
        c.queue = g->d_wide_queue + qword++;
        c.counters = g->d_counters;
        c.topo = plan->topo ? 1 : 0;
        if (c.count == 0) continue;
        const int per_cu_c = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / clique_mma_lds_bytes(cls), (size_t)(2048 / clique_mma_threads(cls))));
        const int cgrid = (int)std::max<long long>(1, std::min<long long>((long long)c.count * (cls == 2 ? 8 : 1), (long long)g->cu_count * per_cu_c));
        HIP_TRY(launch_clique_mma(cls, c, cgrid, stream));
        setup_trace("launch: matrix-core counts");
      }
    }
    if (prof) {
      HIP_TRY(hipStreamSynchronize(stream));
      size_t nt = 0, nh = 0;
      for (const auto &rd : plan->rounds) { nt += rd.n_tasks; nh += rd.host_tab.n; }
      fprintf(stderr, "[clique plan] %zu wide vertices, %lld narrow chunks, %zu tasks in %zu host chunks, %zu round(s), arena %.1f MB, stage %d, %s numbering\n",
              plan->verts.size(), plan->n_count, nt, nh, plan->rounds.size(), g->wide_mat_bytes / 1048576.0, plan->stage, plan->topo ? "topological" : "arbitrary");
    }
  }
  uint64_t chunks_total = (uint64_t)p.count + plan_chunks;
  bool joined[3] = {false, false, false};
  if (use_classes) {
    // A class kernel whose share of chunks cannot fill the chip on its own (small graphs, 1/8 shares) runs on a side stream, so
    // that the kernels launched after it fill the idle CUs; one that can fill it stays on the launch's stream -- there every
    // kernel has the chip to itself (side streams at R-MAT-24 size: diamond 667 vs 636 ms, the 148 KB workgroups of class 2 wait
    // for whole CUs to drain). GM_CLASSES_STREAMS=0 / 1 forces one or the other.
    const char *streams_env = gm_sweep_env("GM_CLASSES_STREAMS");
    auto side_stream = [&](int cls, long long count, long long full_grid, hipStream_t *ws) -> int {
      *ws = stream;
      const bool side = streams_env ? atoi(streams_env) != 0 : count < full_grid;
      if (!side) return GM_OK;
      if (!g->aux_stream[cls - 1]) {
        HIP_TRY(hipStreamCreateWithFlags(&g->aux_stream[cls - 1], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g->aux_done[cls - 1], hipEventDisableTiming));
      }
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
      if (share_rc) return share_rc;
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
    if (share_rc) return share_rc;
    q.g.trp = nullptr;
    q.g.tdesc = nullptr;
    q.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 1;
    if (q.count > 0) {
      chunks_total += (uint64_t)q.count;
      const long long wq = ((long long)q.count + (long long)q.grab - 1) / (long long)q.grab;
      HIP_TRY(launch_mine(pat, q, (int)std::max<long long>(1, std::min<long long>(wq, (long long)g->cu_count * per_cu)), stream));
    }
  }
  if (support) {  // zero the supports, three increments per triangle, then sum C(t, 2): all inside the timed region
    unsigned *sup = sup_part ? sup_out : g->d_sup;
    HIP_TRY(hipMemsetAsync(sup, 0, sizeof(unsigned) * (size_t)(sup_part ? diamond_support_entries(g->ne, world) : g->ne), stream));
    p.scratch = sup;
    if (tab_big) {  // the hosts with rows of 1025 .. 2048 entries first (the heaviest tasks), on the 2048-entry kernel
      MineParams q = p;
      take_share(tab_big, q);
      if (share_rc) return share_rc;
      q.grab = 1;
      q.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 2;
      chunks_total += (uint64_t)q.count;
      if (q.count > 0) HIP_TRY(launch_sup(q, kTctStageMax, (int)std::max<long long>(1, std::min<long long>(q.count, (long long)g->cu_count * sup_per_cu(kTctStageMax))), stream));
    }
    if (p.count > 0) HIP_TRY(launch_sup(p, tct_stage, (int)std::max<long long>(1, std::min<long long>(want, (long long)g->cu_count * sup_per_cu(tct_stage))), stream));
    if (tct_long && g->n_long_rows > 0) {
      SupLongParams sl;
      memset(&sl, 0, sizeof sl);
      sl.rp = g->d_rp;
      sl.col = g->d_col;
      sl.rows = g->d_long_rows;
      sl.prefix = g->d_long_prefix;
      sl.nrows = g->n_long_rows;
      sl.total = g->long_edges;
      sl.sup = sup;
      sl.topo = (g->topo_state == 1 && !gm_sweep_env("GM_TC_NO_TRIM")) ? 1 : 0;
      sl.rank = rank;
      sl.world = world;
      HIP_TRY(launch_sup_long(sl, g->cu_count, stream));
      my_edges += (unsigned long long)((g->long_edges - rank + world - 1) / world);
    }
    if (sup_core) {  // the supports of the hub corner's edges: (A A)_ij on the matrix cores (gm_ctc.hip); a rank takes every world-th block
      CoreTcParams cp;
      memset(&cp, 0, sizeof cp);
      cp.core = g->d_csym;
      cp.h = g->tl_core_h;
      cp.row_words = cp.h / 32;
      cp.first = rank;
      cp.step = world;
      cp.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 4;  // (its own dequeue word inside the zeroed 64-byte block)
      cp.counters = g->d_counters;
      cp.rp = g->d_rp;
      cp.base = corner_from;
      cp.first_pos = g->d_cfirst;
      cp.sup = sup;
      HIP_TRY(hipEventRecord(ctx.evp[2], stream));
      HIP_TRY(launch_core_sup(cp, g->cu_count, stream));
      HIP_TRY(hipEventRecord(ctx.evp[3], stream));
      g->ev_corner[(g->ev_launches - 1) % gm_graph::kEvRing] = true;
    }
    if (sup_masks) {  // the masks of the in-edge tasks, summed by column into the supports of their rows
      SupColsParams sc;
      memset(&sc, 0, sizeof sc);
      sc.nv = g->nv;
      sc.ne = g->ne;
      sc.lmin = sup_mask_min_tail(g);
      sc.rp = g->d_rp;
      sc.emoff = g->d_emoff;
      sc.smask = g->d_smask;
      sc.sup = sup;
      sc.far_rows = g->d_sup_far_rows;
      sc.n_far_rows = g->n_sup_far_rows;
      HIP_TRY(launch_sup_cols(sc, g->cu_count, stream));
    }
    if (!sup_part) HIP_TRY(launch_sup_pairs(sup, 0, g->ne, g->d_counters, g->cu_count, stream));
  } else if (use_tct) {
    if (tab_big) {  // the hosts with rows of 1025 .. 2048 entries first (the heaviest tasks), on the 2048-entry kernel
      MineParams q = p;
      take_share(tab_big, q);
      if (share_rc) return share_rc;
      q.grab = 1;
      q.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 2;
      chunks_total += (uint64_t)q.count;
      if (q.count > 0) HIP_TRY(launch_tch(q, kTctStageMax, (int)std::max<long long>(1, std::min<long long>(q.count, (long long)g->cu_count * tch_per_cu(kTctStageMax))), stream));
    }
    if (p.count > 0) HIP_TRY(launch_tch(p, tct_stage, (int)std::max<long long>(1, std::min<long long>(want, (long long)g->cu_count * tch_per_cu(tct_stage))), stream));
    if (tc_core) {  // the out-edges of the hub corner: one masked bit-matrix product; a rank takes every world-th block
      CoreTcParams cp;
      memset(&cp, 0, sizeof cp);
      cp.core = g->d_core;
      cp.row_words = (g->core_h + 31) / 32;
      cp.row0 = corner_from - g->core_base;
      cp.word0 = cp.row0 >> 5;
      cp.h = g->nv - corner_from;
      cp.ntasks = 0;
      cp.first = rank;
      cp.step = world;
      cp.queue = reinterpret_cast<unsigned *>(g->d_counters + 4) + 4;  // (its own dequeue word inside the zeroed 64-byte block)
      cp.counters = g->d_counters;
      HIP_TRY(hipEventRecord(ctx.evp[2], stream));
      HIP_TRY(launch_core_tc(cp, g->cu_count, stream));
      HIP_TRY(hipEventRecord(ctx.evp[3], stream));
      g->ev_corner[(g->ev_launches - 1) % gm_graph::kEvRing] = true;
      // (gm_stats.tasks: a table's task edges are its rows' entries -- the corner's rows are chunks of the table like any other)
    }
  } else if (p.count > 0) HIP_TRY(launch_mine(pat, p, grid, stream));
#ifdef GM_DEBUG_CHUNKS
  {
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<unsigned long long> ticks(tab->n);
    HIP_TRY(hipMemcpy(ticks.data(), d_ticks, sizeof(unsigned long long) * tab->n, hipMemcpyDeviceToHost));
    dev_free(d_ticks);
    std::vector<ChunkRec> recs(tab->n);
    HIP_TRY(hipMemcpy(recs.data(), tab->d, sizeof(ChunkRec) * tab->n, hipMemcpyDeviceToHost));
    if (int rcv = table_host_views(g, tab)) return rcv;
    std::vector<size_t> idx(tab->n);
    for (size_t i = 0; i < tab->n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ticks[a] > ticks[b]; });
    unsigned long long tot = 0;
    for (auto t : ticks) tot += t;
    if (const char *dump = gm_sweep_env("GM_CHUNK_DUMP")) {
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
  while (g->ring_alias) g = g->ring_alias;  // (a derived handle may itself run on a renumbered copy)
  const unsigned long long have = std::min<unsigned long long>(g->ev_launches, gm_graph::kEvRing);
  const int m = (int)std::min<unsigned long long>((unsigned long long)n, have);
  HIP_TRY(hipSetDevice(g->device));
  for (int i = 0; i < m; ++i) {  // oldest of the last m launches first
    const unsigned long long idx = (g->ev_launches - (unsigned long long)m + (unsigned long long)i) % gm_graph::kEvRing;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev[idx][0], g->ev[idx][1]));
    ms_out[i] = ms;
    // 4-motif: one call = the per-edge kernel on this handle + the rectangle kernel on the renumbered copy + the 4-clique kernel on the
    // cached DAG; the launches of the same recency on those handles belong to the same call
    for (const gm_graph *x : g->ring_extra) {
      const unsigned long long back = (unsigned long long)(m - i);
      while (x && x->ring_alias) x = x->ring_alias;
      if (!x || x->ev_launches < back) continue;
      const unsigned long long xi = (x->ev_launches - back) % gm_graph::kEvRing;
      float xms = 0.f;
      HIP_TRY(hipEventElapsedTime(&xms, x->ev[xi][0], x->ev[xi][1]));
      ms_out[i] += xms;
    }
  }
  *n_out = m;
  return GM_OK;
}

// the part of gm_kernel_times' durations that the hub-corner kernel of a launch took (gm_ctc.hip; 0 for a launch without one): the
// streamed kernels' time is the difference -- what bench.py prices against the HBM roofline
extern "C" int gm_corner_times(const gm_graph *g, int n, double *ms_out, int *n_out) {
  if (!g || !ms_out || !n_out || n < 0) return GM_ERR_INVALID;
  while (g->ring_alias) g = g->ring_alias;
  const unsigned long long have = std::min<unsigned long long>(g->ev_launches, gm_graph::kEvRing);
  const int m = (int)std::min<unsigned long long>((unsigned long long)n, have);
  HIP_TRY(hipSetDevice(g->device));
  for (int i = 0; i < m; ++i) {
    const unsigned long long idx = (g->ev_launches - (unsigned long long)m + (unsigned long long)i) % gm_graph::kEvRing;
    float ms = 0.f;
    if (g->ev_corner[idx]) HIP_TRY(hipEventElapsedTime(&ms, g->ev[idx][2], g->ev[idx][3]));
    ms_out[i] = ms;
  }
  *n_out = m;
  return GM_OK;
}

// The DAG kernels that profit from a TOPOLOGICAL numbering (every edge from a smaller to a larger id: an in-edge task streams only the
// part of N+(u) beyond v, the k-clique matrices are strictly upper triangular) run on the cached renumbered copy of a DAG that is not
// numbered that way (get_relabeled mode 2). tune[6] & 0x200: on the graph as numbered, like the SgL patterns.
// Only where lists are long: the trimmed streams and the triangular counts save work per KEY, the renumbering concentrates the hubs
// in a few host chunks -- measured (profiles/r03/ab_topo_view.txt, TC ms as numbered / renumbered, whole lists / renumbered, trimmed):
// R-MAT-22 6.45 / 6.26 / 5.28, LiveJournal-size flat degrees (5.5 keys per task) 0.88 / 0.85 / 0.86, power law with LiveJournal's
// maximum degree (5.5 keys per task) 1.14 / 1.88 / 1.59. The switch is the mean length of the row a DAG entry sits in, sum d+^2 / |E+|
// (R-MAT-22: 157, R-MAT-22 ef 28: 322, R-MAT-24: 367; the power-law graph: 11.6, flat: 10.5): >= kTopoMinMeanRow -> renumbered. GM_TOPO_MIN_ROW overrides (0: always).
// Round 6: kTopoMinMeanRow = 0 -- with the kernels of rounds 4 - 5 the renumbered copy wins on those two graphs as well (gm_mine.h).
__global__ __launch_bounds__(256) void sum_sq_deg_kernel(int nv, const int *__restrict__ rp, unsigned long long *__restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long s = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
    const unsigned long long d = (unsigned long long)(rp[v + 1] - rp[v]);
    s += d * d;
  }
  s = gm::wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// sum d^2 / ne of a handle's rows = the mean length of the row an entry sits in (cached)
int ensure_mean_sq_deg(gm_graph *self) {
  if (self->mean_sq_deg < 0) {
    HIP_TRY(hipSetDevice(self->device));
    unsigned long long *d_s = nullptr, s2 = 0;
    HIP_TRY(dev_malloc(&d_s, 8));
    hipError_t e = hipMemset(d_s, 0, 8);
    if (e == hipSuccess && self->nv > 0)
      hipLaunchKernelGGL(sum_sq_deg_kernel, dim3((unsigned)std::min<long long>(((long long)self->nv + 255) / 256, 2048)), dim3(256), 0, 0, self->nv, self->d_rp, d_s);
    if (e == hipSuccess) e = hipMemcpy(&s2, d_s, 8, hipMemcpyDeviceToHost);
    dev_free(d_s);
    if (e != hipSuccess) return hip_fail(e, "sum_sq_deg_kernel", __FILE__, __LINE__);
    self->mean_sq_deg = self->ne > 0 ? (double)s2 / (double)self->ne : 0.0;
  }
  return GM_OK;
}

// Triangles per DAG entry, ESTIMATED: every stride-th entry (u -> v), |N+(u) ^ N+(v)| by a merge of the two sorted rows (at most 512 steps:
// the estimate is asked for on short-row graphs), averaged.  The edge supports choose between the key stream and the task lists by it.
__global__ __launch_bounds__(256) void tri_sample_kernel(int nv, long long ne, const int *__restrict__ rp, const int *__restrict__ col, long long stride,
                                                         int nsamples, unsigned long long *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  if (i < nsamples) {
    const long long e = (long long)i * stride;
    if (e < ne) {
      int lo = 0, hi = nv - 1;  // the row of entry e
      while (lo < hi) {
        const int mid = (int)(((long long)lo + hi + 1) >> 1);
        if ((long long)rp[mid] <= e) lo = mid; else hi = mid - 1;
      }
      const int v = col[e];
      int a = rp[lo], ae = rp[lo + 1], b = rp[v], be = rp[v + 1];
      for (int step = 0; step < 512 && a < ae && b < be; ++step) {
        const int x = col[a], y = col[b];
        c += x == y ? 1ull : 0ull;
        a += x <= y ? 1 : 0;
        b += y <= x ? 1 : 0;
      }
    }
  }
  c = gm::wave_sum_u64(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int ensure_tri_per_edge(gm_graph *self) {
  if (self->tri_per_edge < 0) {
    HIP_TRY(hipSetDevice(self->device));
    const int nsamples = (int)std::min<long long>(1 << 16, std::max<long long>(self->ne, 1));
    const long long stride = std::max<long long>(1, self->ne / nsamples);
    unsigned long long *d_s = nullptr, s = 0;
    HIP_TRY(dev_malloc(&d_s, 8));
    hipError_t e = hipMemset(d_s, 0, 8);
    if (e == hipSuccess && self->ne > 0)
      hipLaunchKernelGGL(tri_sample_kernel, dim3((unsigned)((nsamples + 255) / 256)), dim3(256), 0, 0, self->nv, (long long)self->ne, self->d_rp, self->d_col, stride,
                         nsamples, d_s);
    if (e == hipSuccess) e = hipMemcpy(&s, d_s, 8, hipMemcpyDeviceToHost);
    dev_free(d_s);
    if (e != hipSuccess) return hip_fail(e, "tri_sample_kernel", __FILE__, __LINE__);
    self->tri_per_edge = (double)s / (double)nsamples;
  }
  return GM_OK;
}

// Heavy chunks of the task-list kernels (a hub hosts 10^5 in-edges) are cut into PARTS that share the chunk's batches -- and every part
// hashes the chunk's rows again (~10 us).  A part bounds the tail of a rank's launch, so its size follows the rank's share: the
// launch streams about ne * (sum d+^2 / ne) / 2 keys, a part takes 1/3000 of a rank's share of them, between 128 K and 4 M keys.
// One-GPU simulation of 8 rank shares, TC, parts of 64 K / 128 K / 256 K / 512 K / 1 M keys: R-MAT-24 9.08 / 7.14 / 6.11 / 5.68 / 5.54 ms per
// rank (ideal 5.25), R-MAT-22 0.63 / 0.56 / 0.55 / 0.55 / 0.78; one GPU, 256 K / 512 K / 1 M / 2 M / 4 M: R-MAT-22 3.14 / 3.09 / 3.07 / 3.06 / 3.04 ms,
// R-MAT-24 47.2 / 43.7 / 41.9 / 41.2 / 40.9 (profiles/r03/ab_share_scaling.txt).
unsigned long long task_part_cap(gm_graph *g, int world) {
  if (const char *e = gm_sweep_env("GM_TCT_PART_KKEYS")) return (unsigned long long)std::max(4, atoi(e)) << 10;  // (sweeps)
  double keys = 0.0;
  if (ensure_mean_sq_deg(g) == GM_OK) keys = (double)g->ne * g->mean_sq_deg * 0.5;
  const double cap = keys / (3000.0 * (double)std::max(world, 1));
  return (unsigned long long)std::min(std::max(cap, (double)(128 << 10)), (double)(4 << 20));
}

static int topo_view(const gm_graph *dag, const gm_launch *la, gm_graph **run_on) {
  gm_graph *self = const_cast<gm_graph *>(dag);
  if (!self) return GM_ERR_INVALID;
  if (int rc = reject_big(self)) return rc;
  *run_on = self;
  if (la && (la->tune[6] & 0x200)) return GM_OK;
  bool topo = false;
  int rc = graph_is_topological(self, &topo);
  if (rc || topo) return rc;
  rc = ensure_mean_sq_deg(self);
  if (rc) return rc;
  double min_row = (double)kTopoMinMeanRow;
  if (const char *e = gm_opt("GM_TOPO_MIN_ROW")) min_row = atof(e);
  if (gm_sweep_env("GM_TABLE_INFO")) fprintf(stderr, "[topo view] sum d+^2 / |E+| = %.1f (switch at %.1f)\n", self->mean_sq_deg, min_row);
  if (self->mean_sq_deg < min_row) return GM_OK;  // short lists: as numbered
  return get_relabeled(self, 2, run_on);
}

static int run_tc(const gm_graph *dag, const gm_launch *la, uint64_t *out, int nout, gm_stats *st, int fin_mode = -1, unsigned long long fin_base = 0) {
  gm_graph *run_on = nullptr;
  int rc = topo_view(dag, la, &run_on);
  if (rc) return rc;
  rc = run_pattern(PAT_TC, run_on, la, 3, out, nout, st, fin_mode, fin_base);
  const_cast<gm_graph *>(dag)->ring_alias = (run_on != dag) ? run_on : nullptr;
  return rc;
}

extern "C" int gm_tc(const gm_graph *dag, const gm_launch *la, uint64_t *total, gm_stats *st) { return run_tc(dag, la, total, 1, st); }
// tooling (the byte model of bench.py, tests): the hub corner the triangle count of this handle takes on the matrix cores (after a first gm_tc)
extern "C" int gm_tc_core_info(const gm_graph *dag, int64_t info[4]) {
  if (!dag || !info) return GM_ERR_INVALID;
  gm_graph *run_on = nullptr;
  // (a symmetric handle: the oriented copy its formula 3-motif counts the triangles of)
  const int rc = topo_view(dag->dag_cache ? dag->dag_cache : dag, nullptr, &run_on);
  if (rc) return rc;
  const bool on = run_on->kst_skip_from < run_on->nv;
  const long long nJ = on ? (run_on->tc_core_h + 63) >> 6 : 0;
  info[0] = on ? run_on->tc_core_h : 0;           // vertices of the corner (0: every edge is a task of the key stream)
  info[1] = on ? run_on->tc_core_edges : 0;       // DAG entries inside it
  info[2] = nJ * (nJ + 1) / 2;                    // 64 x 64 blocks of the masked product
  info[3] = on ? run_on->core_h : 0;              // vertices of the core bitmap the corner is a part of
  return GM_OK;
}

// rectangle, flattened over wedges (rect_flat_kernel in gm_mine.hip)
static int ensure_idx0(gm_graph *g, const GraphView &gv) {
  if (g->d_idx0) return GM_OK;
  OtherSetupScope scope(g);
  HIP_TRY(dev_malloc(&g->d_idx0, sizeof(int) * (size_t)std::max(g->nv, 1)));
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
    HIP_TRY(dev_malloc(&g->d_wblock_prefix, sizeof(unsigned long long) * ((size_t)g->nv + 1)));
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
  const bool lds_maps = !pentagon && !(la->tune[6] & 0x20000);
  if (!g->d_rect_tasks && !lds_maps) {  // once per graph: 2-path estimate per centre (device), task list (host): heavy first, then light by 4
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long *d_work = nullptr;
    HIP_TRY(dev_malloc(&d_work, sizeof(unsigned long long) * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1));
    hipError_t e = nv ? launch_rect_work(gv, g->d_idx0, d_work, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
    dev_free(d_work);
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
    HIP_TRY(dev_malloc(&g->d_rect_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_rect_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
  }
  // rectangle: the 2-path ends in the last kRectLdsRanges * kRectLdsRange ids of the centres with >= GM_RECT_LDS_MIN 2-paths are counted in
  // LDS maps (rect_lds_kernel); tune[6] & 0x20000: every end in the global maps, round 5's form.  Once per graph: the row bounds, the
  // (centre, range) tasks -- every centre's own top range first, the centres heaviest first -- and rect_acc_kernel's list again with
  // those centres (now only their ends below the cut) in front.
  setup_trace("rect: idx0 / old tasks");
  if (lds_maps && !g->rect_lds_ready) {
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long lds_min = 4096;
    if (const char *e = gm_opt("GM_RECT_LDS_MIN")) lds_min = std::strtoull(e, nullptr, 10);
    // the ranges, from the last ids down (the hubs of a graph numbered ascending in degree): a range's counters are as wide as the largest
    // degree among its vertices needs -- whatever the numbering (tune[6] & 512 runs on the graph as given)
    RectLdsRanges &rr = g->rect_ranges;
    memset(&rr, 0, sizeof rr);
    {
      const int nblk = (int)((nv + kRectLdsWords - 1) / kRectLdsWords);
      std::vector<int> bmax((size_t)std::max(nblk, 1), 0);
      if (nv) {
        int *d_bmax = nullptr;
        HIP_TRY(dev_malloc(&d_bmax, sizeof(int) * (size_t)nblk));
        hipError_t e0 = hipMemset(d_bmax, 0, sizeof(int) * (size_t)nblk);
        if (e0 == hipSuccess) e0 = launch_rect_blockmax(gv, d_bmax, 0);
        if (e0 == hipSuccess) e0 = hipMemcpy(bmax.data(), d_bmax, sizeof(int) * (size_t)nblk, hipMemcpyDeviceToHost);
        dev_free(d_bmax);
        if (e0 != hipSuccess) return hip_fail(e0, "rect_blockmax_kernel", __FILE__, __LINE__);
      }
      auto max_deg_of_blocks = [&](int b0, int nb) {  // blocks b0 .. b0 + nb - 1 (counted from the top), those that exist
        int m = 0;
        for (int b = b0; b < std::min(b0 + nb, nblk); ++b) m = std::max(m, bmax[(size_t)b]);
        return m;
      };
      int blk = 0;
      int top[kRectLdsRanges + 1], lbs[kRectLdsRanges], n = 0;
      long long hi = (long long)g->nv;
      top[0] = (int)hi;
      int max_ranges = kRectLdsRanges;
      if (const char *e = gm_opt("GM_RECT_LDS_RANGES")) max_ranges = std::max(1, std::min(kRectLdsRanges, std::atoi(e)));
      while (hi > 0 && n < max_ranges) {
        const int lb = max_deg_of_blocks(blk, 4) < 256 ? 3 : max_deg_of_blocks(blk, 2) < 65536 ? 4 : 5;
        const long long width = (long long)kRectLdsWords << (5 - lb);
        blk += 1 << (5 - lb);
        hi = std::max<long long>(0, hi - width);
        lbs[n] = lb;
        top[++n] = (int)hi;
      }
      rr.n = n;
      for (int k = 0; k <= n; ++k) rr.rb[k] = top[n - k];
      for (int k = 0; k < n; ++k) rr.lb[k] = lbs[n - 1 - k];
    }
    setup_trace("rect: ranges");
    g->rect_cut = rr.rb[0];
    HIP_TRY(dev_malloc(&g->d_rect_bnd, sizeof(int) * (size_t)(rr.n + 1) * std::max<size_t>(nv, 1)));
    unsigned long long *d_work = nullptr;
    HIP_TRY(dev_malloc(&d_work, sizeof(unsigned long long) * 2 * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1)), wcut(std::max<size_t>(nv, 1));
    hipError_t e = hipSuccess;
    if (nv) {
      e = launch_rect_bounds(gv, rr, g->d_rect_bnd, 0);
      if (e == hipSuccess) e = launch_rect_work_cut(gv, g->d_idx0, g->d_rect_bnd, rr.n + 1, d_work, 0);  // (both estimates: [0, nv) all ends, [nv, 2 nv) the ends below the cut)
      if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(wcut.data(), d_work + nv, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
    }
    dev_free(d_work);
    if (e != hipSuccess) return hip_fail(e, "rect_bounds_kernel", __FILE__, __LINE__);
    setup_trace("rect: bounds + work estimates");
    std::vector<int> lds, rest;
    for (size_t v = 0; v < nv; ++v) {
      if (work[v] == 0) continue;
      if (work[v] >= lds_min && (long long)v > (long long)g->rect_cut) lds.push_back((int)v);
      else rest.push_back((int)v);
    }
    auto by = [](const std::vector<unsigned long long> &w) { return [&w](int a, int b) { return w[(size_t)a] > w[(size_t)b]; }; };
    std::stable_sort(lds.begin(), lds.end(), by(work));
    std::stable_sort(rest.begin(), rest.end(), by(work));
    std::vector<int2> lt;
    auto range_of = [&](int w) {  // the range that holds id w >= cut
      int k = 0;
      while (k + 1 < rr.n && rr.rb[k + 1] <= w) ++k;
      return k;
    };
    // a centre with more neighbours below it than rect_lds_kernel has threads: a task per range, every centre's own top range first; the
    // others: one task for all their ranges
    std::vector<int> idx0h(std::max<size_t>(nv, 1));
    if (nv) HIP_TRY(hipMemcpy(idx0h.data(), g->d_idx0, sizeof(int) * nv, hipMemcpyDeviceToHost));
    const int per_wg = kRectLdsWaves * GM_WAVE;
    for (int j = 0; j < rr.n; ++j)
      for (int v : lds) {
        if (idx0h[(size_t)v] <= per_wg) continue;
        const int k = range_of(v - 1) - j;
        if (k >= 0) lt.push_back(make_int2(v, k));
      }
    for (int v : lds)
      if (idx0h[(size_t)v] <= per_wg) lt.push_back(make_int2(v, -1));
    const unsigned long long heavy = 1ull << 15;  // 2-paths above which a centre gets a whole workgroup of rect_acc_kernel
    std::vector<int4> tasks;
    auto emit = [&](const std::vector<int> &vs, const std::vector<unsigned long long> &w) {
      size_t i = 0;
      for (; i < vs.size() && w[(size_t)vs[i]] >= heavy; ++i) tasks.push_back(make_int4(vs[i], -2, -2, -2));
      for (; i < vs.size(); i += 4) {
        int4 t = make_int4(vs[i], -1, -1, -1);
        if (i + 1 < vs.size()) t.y = vs[i + 1];
        if (i + 2 < vs.size()) t.z = vs[i + 2];
        if (i + 3 < vs.size()) t.w = vs[i + 3];
        tasks.push_back(t);
      }
    };
    std::vector<int> ldscut;  // the LDS centres that have ends below the cut (none when the ranges cover the whole graph)
    for (int v : lds)
      if (wcut[(size_t)v] > 0) ldscut.push_back(v);
    std::stable_sort(ldscut.begin(), ldscut.end(), by(wcut));
    emit(ldscut, wcut);
    g->n_rect_cut = tasks.size();
    emit(rest, work);
    g->n_rect_cut_tasks = tasks.size();
    g->n_rect_lds_tasks = lt.size();
    HIP_TRY(dev_malloc(&g->d_rect_cut_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_rect_cut_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
    HIP_TRY(dev_malloc(&g->d_rect_lds_tasks, sizeof(int2) * std::max<size_t>(lt.size(), 1)));
    if (!lt.empty()) HIP_TRY(hipMemcpy(g->d_rect_lds_tasks, lt.data(), sizeof(int2) * lt.size(), hipMemcpyHostToDevice));
    g->rect_lds_ready = true;
    setup_trace("rect: task lists");
  }
  RectAccParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.idx0 = g->d_idx0;
  p.tasks = lds_maps ? g->d_rect_cut_tasks : g->d_rect_tasks;
  if (lds_maps) {
    p.n_cut = g->n_rect_cut;
    p.cut = g->rect_cut;
    p.bnd0 = g->d_rect_bnd;
    p.bnd_stride = g->rect_ranges.n + 1;
  }
  int64_t first = 0, step = 1, count = 0;
  gm_partition((int64_t)(lds_maps ? g->n_rect_cut_tasks : g->n_rect_tasks), ctx.rank, ctx.world, la->policy, &first, &step, &count);
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
  // (with the heavy centres in LDS what is left for the global maps is light: 1 / 2 / 4 / 8 workgroups per CU measured 11.15 / 10.95 /
  // 11.1 / 11.0 ms for the rectangle of R-MAT-20 -- two, so that a first call does not allocate 2 x 32 GB of maps and lists)
  int acc_wgs_per_cu = lds_maps ? 2 : 8;
  if (const char *e = gm_sweep_env("GM_RECT_ACC_WGS")) acc_wgs_per_cu = std::max(1, std::atoi(e));
  long long grid = std::min<long long>((long long)g->cu_count * acc_wgs_per_cu, (long long)std::max<unsigned long long>(1, budget / std::max<unsigned long long>(per_wg, 1)));
  grid = std::max<long long>(1, std::min<long long>(grid, count));
  const size_t need = (size_t)per_wg * (size_t)grid;
  if (need > g->rect_acc_bytes) {
    if (g->d_rect_acc) dev_free(g->d_rect_acc);
    g->d_rect_acc = nullptr;
    g->rect_acc_bytes = 0;
    HIP_TRY(dev_malloc(&g->d_rect_acc, need));
    HIP_TRY(hipMemset(g->d_rect_acc, 0, need));  // every launch leaves the maps zeroed again
    g->rect_acc_bytes = need;
  }
  p.acc = g->d_rect_acc;
  if (need > g->pent_touched_bytes) {  // touched-vertex lists: same shape as the maps (one int list of up to nv entries per wave)
    if (g->d_pent_touched) dev_free(g->d_pent_touched);
    g->d_pent_touched = nullptr;
    g->pent_touched_bytes = 0;
    HIP_TRY(dev_malloc(&g->d_pent_touched, need));
    g->pent_touched_bytes = need;
  }
  p.touched = g->d_pent_touched;
  setup_trace("rect: global maps");
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
  RectLdsParams lp;
  memset(&lp, 0, sizeof lp);
  int64_t lcount = 0;
  if (lds_maps) {
    lp.g = gv;
    lp.idx0 = g->d_idx0;
    lp.tasks = g->d_rect_lds_tasks;
    int64_t lfirst = 0, lstep = 1;
    gm_partition((int64_t)g->n_rect_lds_tasks, ctx.rank, ctx.world, la->policy, &lfirst, &lstep, &lcount);
    lp.first = (unsigned long long)lfirst;
    lp.step = (unsigned long long)lstep;
    lp.count = (unsigned long long)lcount;
    lp.bnd = g->d_rect_bnd;
    lp.r = g->rect_ranges;
    lp.queue = g->d_counters + 5;  // (its own dequeue word inside the zeroed 64-byte block)
    lp.counters = g->d_counters;
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  // (workgroups per CU: what its LDS -- the map + 1.5 KB of scratch per wave -- lets run together)
  const int lds_wgs = std::max(1, (int)((160 * 1024) / (kRectLdsWords * 4 + kRectLdsWaves * 1536 + 1024)));
  if (lcount > 0) HIP_TRY(launch_rect_lds(lp, (int)std::min<long long>((long long)g->cu_count * lds_wgs, (long long)lcount), ctx.stream));
  setup_trace("rect: lds kernel enqueued");
  if (count > 0) HIP_TRY(launch_rect_acc(p, (int)grid, ctx.stream));
  setup_trace("rect: acc kernel enqueued");
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)(count + lcount), (int)grid, 256);
  return end_launch(ctx, FIN_COPY, 0, h_out, 1, st);
}

// per-entry triangle tables t / tlt (edge_tab_kernel), once per graph; every rank builds the whole tables: they are inputs of
// every centre of the house / pentagon map kernels
static int ensure_edge_tables(gm_graph *g, const GraphView &gv) {
  if (g->d_house_t && g->d_house_tlt) return GM_OK;
  OtherSetupScope scope(g);
  const size_t ne1 = (size_t)std::max<long long>(g->ne, 1);
  DevBuf<unsigned> t, tlt;
  HIP_TRY(t.alloc(ne1, true));  // (handed to the handle below)
  HIP_TRY(tlt.alloc(ne1, true));
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
  // the maps of the centres with >= GM_RECT_LDS_MIN 2-paths in LDS, range by range (house_lds_kernel); tune[6] & 0x20000: every end in the
  // global maps, round 5's form.  Once per graph: ranges of kHouseLdsIds ids from the last id down, the row bounds, the tasks.
  const bool lds_maps = !(la->tune[6] & 0x20000);
  if (lds_maps && !g->house_lds_ready) {
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long lds_min = 4096;
    if (const char *e = gm_opt("GM_RECT_LDS_MIN")) lds_min = std::strtoull(e, nullptr, 10);
    int max_ranges = kHouseLdsRanges;
    if (const char *e = gm_opt("GM_RECT_LDS_RANGES")) max_ranges = std::max(1, std::min(kHouseLdsRanges, std::atoi(e)));
    HouseLdsRanges &rr = g->house_ranges;
    memset(&rr, 0, sizeof rr);
    // (the row-bound table has n + 1 entries per vertex: at most 4 GB of it)
    const long long by_memory = std::max<long long>(1, (4ll << 30) / (4ll * (long long)std::max<size_t>(nv, 1)) - 1);
    rr.n = (int)std::min<long long>(std::min<long long>(max_ranges, by_memory), ((long long)g->nv + kHouseLdsIds - 1) / kHouseLdsIds);
    rr.nv = g->nv;
    rr.cut = (int)std::max<long long>(0, (long long)g->nv - (long long)rr.n * kHouseLdsIds);
    HIP_TRY(dev_malloc(&g->d_house_bnd, sizeof(int) * (size_t)(rr.n + 1) * std::max<size_t>(nv, 1)));
    unsigned long long *d_work = nullptr;
    HIP_TRY(dev_malloc(&d_work, sizeof(unsigned long long) * 2 * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1)), wcut(std::max<size_t>(nv, 1));
    std::vector<int> rph(nv + 1, 0);
    hipError_t e = hipSuccess;
    if (nv) {
      e = launch_house_bounds(gv, rr, g->d_house_bnd, 0);
      if (e == hipSuccess) e = launch_house_work_cut(gv, g->d_house_bnd, rr.n + 1, d_work, 0);
      if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(wcut.data(), d_work + nv, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipMemcpy(rph.data(), g->d_rp, sizeof(int) * (nv + 1), hipMemcpyDeviceToHost);
    }
    dev_free(d_work);
    if (e != hipSuccess) return hip_fail(e, "house_bounds_kernel", __FILE__, __LINE__);
    // A centre with at most one neighbour per thread is ONE task that walks every range -- two per walk -- at ~2.5 us of a CU per walk
    // whatever it finds there, against ~12 ns per 2-path in the global maps (both measured on R-MAT-20): it pays from ~200 2-paths per
    // walk on.  (On a LiveJournal-sized power-law graph -- 4.8 M vertices, 111 walks -- the fixed 4096 made the house 1.27 x SLOWER than
    // the global maps.)  A developer option that sets the threshold is taken as given.
    const bool lds_min_given = gm_opt("GM_RECT_LDS_MIN") != nullptr;
    unsigned long long per_walk = 200;
    if (const char *e = gm_sweep_env("GM_HOUSE_WALK_MIN")) per_walk = std::strtoull(e, nullptr, 10);
    auto lds_min_of = [&](int v) {
      const int d = rph[(size_t)v + 1] - rph[(size_t)v];
      if (lds_min_given || d > kRectLdsWaves * GM_WAVE) return lds_min;
      const int per = d <= 32 ? 4 : 2;  // ranges per walk (house_lds_kernel: 16-bit counters up to 32 neighbours)
      return std::max<unsigned long long>(lds_min, per_walk * (unsigned long long)((rr.n + per - 1) / per));
    };
    std::vector<int> lds, rest;
    for (size_t v = 0; v < nv; ++v) {
      if (work[v] == 0) continue;
      if (rr.n > 0 && work[v] >= lds_min_of((int)v)) lds.push_back((int)v);
      else rest.push_back((int)v);
    }
    auto by = [](const std::vector<unsigned long long> &w) { return [&w](int a, int b) { return w[(size_t)a] > w[(size_t)b]; }; };
    std::stable_sort(lds.begin(), lds.end(), by(work));
    std::stable_sort(rest.begin(), rest.end(), by(work));
    const int per_wg = kRectLdsWaves * GM_WAVE;
    auto deg = [&](int v) { return rph[(size_t)v + 1] - rph[(size_t)v]; };
    std::vector<int2> lt;
    for (int k = rr.n - 1; k >= 0; --k)  // (the hubs' ranges first: they hold most ends)
      for (int v : lds)
        if (deg(v) > per_wg) lt.push_back(make_int2(v, k));
    for (int v : lds)
      if (deg(v) <= per_wg) lt.push_back(make_int2(v, -1));
    const unsigned long long heavy = 1ull << 15;
    std::vector<int4> tasks;
    auto emit = [&](const std::vector<int> &vs, const std::vector<unsigned long long> &w) {
      size_t i = 0;
      for (; i < vs.size() && w[(size_t)vs[i]] >= heavy; ++i) tasks.push_back(make_int4(vs[i], -2, -2, -2));
      for (; i < vs.size(); i += 4) {
        int4 t4 = make_int4(vs[i], -1, -1, -1);
        if (i + 1 < vs.size()) t4.y = vs[i + 1];
        if (i + 2 < vs.size()) t4.z = vs[i + 2];
        if (i + 3 < vs.size()) t4.w = vs[i + 3];
        tasks.push_back(t4);
      }
    };
    // (every LDS centre stays in house_acc_kernel's list: its phase 0 -- the table terms and the intersections -- is done there, and of its
    // 2-paths the ends below the cut; ordered by that remainder, the intersections taken as its degree)
    std::vector<unsigned long long> wrem(std::max<size_t>(nv, 1), 0);
    for (int v : lds) wrem[(size_t)v] = wcut[(size_t)v] + (unsigned long long)deg(v);
    std::stable_sort(lds.begin(), lds.end(), by(wrem));
    emit(lds, wrem);
    g->n_house_cut = tasks.size();
    emit(rest, work);
    g->n_house_cut_tasks = tasks.size();
    g->n_house_lds_tasks = lt.size();
    HIP_TRY(dev_malloc(&g->d_house_cut_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_house_cut_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
    HIP_TRY(dev_malloc(&g->d_house_lds_tasks, sizeof(int2) * std::max<size_t>(lt.size(), 1)));
    if (!lt.empty()) HIP_TRY(hipMemcpy(g->d_house_lds_tasks, lt.data(), sizeof(int2) * lt.size(), hipMemcpyHostToDevice));
    g->house_lds_ready = true;
  }
  if (!g->d_house_tasks && !lds_maps) {  // once per graph
    OtherSetupScope scope(g);
    const size_t nv = (size_t)g->nv;
    unsigned long long *d_work = nullptr;
    HIP_TRY(dev_malloc(&d_work, sizeof(unsigned long long) * std::max<size_t>(nv, 1)));
    std::vector<unsigned long long> work(std::max<size_t>(nv, 1));
    hipError_t e = nv ? launch_house_work(gv, d_work, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(work.data(), d_work, sizeof(unsigned long long) * nv, hipMemcpyDeviceToHost);
    dev_free(d_work);
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
    HIP_TRY(dev_malloc(&g->d_house_tasks, sizeof(int4) * std::max<size_t>(tasks.size(), 1)));
    if (!tasks.empty()) HIP_TRY(hipMemcpy(g->d_house_tasks, tasks.data(), sizeof(int4) * tasks.size(), hipMemcpyHostToDevice));
  }
  HouseAccParams p;
  memset(&p, 0, sizeof p);
  p.g = gv;
  p.t = g->d_house_t;
  p.tlt = g->d_house_tlt;
  p.tasks = lds_maps ? g->d_house_cut_tasks : g->d_house_tasks;
  if (lds_maps) {
    p.n_cut = g->n_house_cut;
    p.cut = g->house_ranges.cut;
    p.bnd0 = g->d_house_bnd;
    p.bnd_stride = g->house_ranges.n + 1;
  }
  int64_t first = 0, step = 1, count = 0;
  gm_partition((int64_t)(lds_maps ? g->n_house_cut_tasks : g->n_house_tasks), ctx.rank, ctx.world, la->policy, &first, &step, &count);
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
  long long grid = std::min<long long>((long long)g->cu_count * (lds_maps ? 4 : 8), (long long)std::max<unsigned long long>(1, budget / std::max<unsigned long long>(per_wg, 1)));
  grid = std::max<long long>(1, std::min<long long>(grid, count));
  const size_t need = (size_t)per_wg * (size_t)grid;
  if (need > g->house_acc_bytes) {
    if (g->d_house_acc) dev_free(g->d_house_acc);
    g->d_house_acc = nullptr;
    g->house_acc_bytes = 0;
    HIP_TRY(dev_malloc(&g->d_house_acc, need));
    HIP_TRY(hipMemset(g->d_house_acc, 0, need));  // every launch leaves the maps zeroed again
    if (g->d_house_touched) dev_free(g->d_house_touched);
    g->d_house_touched = nullptr;
    HIP_TRY(dev_malloc(&g->d_house_touched, need / 2));  // one int list per map
    g->house_acc_bytes = need;
  }
  p.acc = g->d_house_acc;
  p.touched = g->d_house_touched;
  HouseLdsParams lp;
  memset(&lp, 0, sizeof lp);
  int64_t lcount = 0;
  if (lds_maps) {
    lp.g = gv;
    lp.t = g->d_house_t;
    lp.tasks = g->d_house_lds_tasks;
    int64_t lfirst = 0, lstep = 1;
    gm_partition((int64_t)g->n_house_lds_tasks, ctx.rank, ctx.world, la->policy, &lfirst, &lstep, &lcount);
    lp.first = (unsigned long long)lfirst;
    lp.step = (unsigned long long)lstep;
    lp.count = (unsigned long long)lcount;
    lp.bnd = g->d_house_bnd;
    lp.r = g->house_ranges;
    lp.queue = g->d_counters + 5;  // (its own dequeue word inside the zeroed 64-byte block)
    lp.counters = g->d_counters;
  }
  rc = start_timer(ctx);
  if (rc) return rc;
  if (lcount > 0) HIP_TRY(launch_house_lds(lp, (int)std::min<long long>((long long)g->cu_count, (long long)lcount), ctx.stream));
  if (count > 0) HIP_TRY(launch_house_acc(p, (int)grid, ctx.stream));
  fill_stats(st, (uint64_t)(g->ne / 2 / ctx.world), (uint64_t)(count + lcount), (int)grid, 256);
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
    HIP_TRY(dev_malloc(&d_nblk, sizeof(unsigned) * std::max<size_t>(ne, 1)));
    std::vector<unsigned> nblk(std::max<size_t>(ne, 1));
    hipError_t e = ne ? launch_house_blocks(gv, d_nblk, 0) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpy(nblk.data(), d_nblk, sizeof(unsigned) * ne, hipMemcpyDeviceToHost);
    dev_free(d_nblk);
    if (e != hipSuccess) return hip_fail(e, "house block table", __FILE__, __LINE__);
    std::vector<unsigned long long> pre(ne + 1);
    unsigned long long acc = 0;
    for (size_t i = 0; i < ne; ++i) { pre[i] = acc; acc += nblk[i]; }
    pre[ne] = acc;
    g->n_house_blocks = acc;
    HIP_TRY(dev_malloc(&g->d_house_prefix, sizeof(unsigned long long) * (ne + 1)));
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
      if (g->d_scratch) dev_free(g->d_scratch);
      g->d_scratch = nullptr;
      g->scratch_bytes = 0;
      HIP_TRY(dev_malloc(&g->d_scratch, need));
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

// The oriented copy of a symmetric handle (gm_graph_orient), built once under its own lock: two solvers called on the same handle from two
// threads -- the formula 3-motif and the one-GPU diamond share it -- must not both orient and leak a handle.  A graph of
// >= 2^31 entries too: its oriented copy must fit the 32-bit task index, gm_graph_orient checks.
static int ensure_dag_cache(gm_graph *g) {
  std::lock_guard<std::mutex> lk(g->dag_mu);
  if (g->dag_cache) return GM_OK;
  gm_graph *dag = nullptr;
  const int rc = gm_graph_orient(g, &dag);
  if (rc) return rc;
  dag->pool_owner = g;
  g->dag_cache = dag;
  return GM_OK;
}

// diamond on one GPU: the edge supports of the oriented copy (cached on the handle; on its topological view where lists are long)
static int run_diamond_supports(const gm_graph *sym, const gm_launch *la, uint64_t *total, gm_stats *st) {
  gm_graph *g = const_cast<gm_graph *>(sym);
  if (!sym) return GM_ERR_INVALID;
  if (la && la->world > 1) return GM_ERR_UNSUPPORTED;
  {  // (the oriented copy, cached on the handle)
    const int rc_dag = ensure_dag_cache(g);
    if (rc_dag) return rc_dag;
  }
  gm_graph *dag = g->dag_cache, *run_on = nullptr;
  int rc = topo_view(dag, la, &run_on);
  if (rc) return rc;
  rc = run_pattern(PAT_SUPPORT, run_on, la, 3, total, 1, st);
  if (rc) return rc;
  dag->ring_alias = (run_on != dag) ? run_on : nullptr;
  g->ring_alias = dag;
  g->ring_extra[0] = g->ring_extra[1] = nullptr;
  return GM_OK;
}

// ---- diamond on several ranks: ONE shared pass over the triangles (the one-GPU algorithm at every N) --------------------------------------
// Every rank runs its share of the triangle pass into its own zeroed support array (one uint32 per entry of the oriented copy, padded so
// that every rank gets the same number of entries), the arrays are summed by ONE reduce-scatter over xGMI (ncclReduceScatter, ncclUint32,
// ncclSum: rank r receives the entries [r n / world, (r + 1) n / world)), every rank takes sum C(t, 2) of its slice, and the 64-bit
// counts meet in the usual all-reduce.  The reference has no multi-GPU diamond at all (src/sgl/multigpu.cu:117 is commented out).
static int diamond_run_on(const gm_graph *sym, const gm_launch *la, gm_graph **run_on) {
  gm_graph *g = const_cast<gm_graph *>(sym);
  if (!sym) return GM_ERR_INVALID;
  {  // (the oriented copy, cached on the handle)
    const int rc_dag = ensure_dag_cache(g);
    if (rc_dag) return rc_dag;
  }
  gm_graph *dag = g->dag_cache;
  return topo_view(dag, la, run_on);  // (rows beyond the 2048-entry stage: their out-edges through sup_long_kernel)
}
extern "C" int gm_diamond_support_size(const gm_graph *sym, int world, int64_t *n_entries) {
  if (!sym || !n_entries || world < 1) return GM_ERR_INVALID;
  gm_graph *run_on = nullptr;
  const int rc = diamond_run_on(sym, nullptr, &run_on);
  if (rc) return rc;
  *n_entries = diamond_support_entries(run_on->ne, world);
  return GM_OK;
}
// tooling (the byte model of bench.py, tests): what the most recent one-GPU diamond of this handle did with its streamed edges
extern "C" int gm_diamond_support_info(const gm_graph *sym, int64_t info[4]) {
  if (!sym || !info) return GM_ERR_INVALID;
  gm_graph *run_on = nullptr;
  const int rc = diamond_run_on(sym, nullptr, &run_on);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(run_on->device));
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long at = 0;
  HIP_TRY(hipMemcpy(&at, run_on->d_counters + 2, sizeof at, hipMemcpyDeviceToHost));
  info[0] = run_on->smask_state == 1 ? (int64_t)run_on->smask_words : 0;  // 64-bit words of the match-mask arena (0: no masks)
  info[1] = (int64_t)at;                                                  // increments issued as global atomics from the waves' queues
  info[2] = run_on->smask_state == 1 ? (int64_t)run_on->n_sup_far_rows : 0;
  info[3] = (int64_t)sup_mask_min_tail(run_on);
  return GM_OK;
}
// tooling: the hub corner whose supports the one-GPU diamond of this handle takes on the matrix cores (after a first diamond)
extern "C" int gm_sup_core_info(const gm_graph *sym, int64_t info[4]) {
  if (!sym || !info) return GM_ERR_INVALID;
  gm_graph *run_on = nullptr;
  const int rc = diamond_run_on(sym, nullptr, &run_on);
  if (rc) return rc;
  const bool on = run_on->tl_skip_from < run_on->nv;
  const long long nb = on ? run_on->tl_core_h >> 8 : 0;
  info[0] = on ? run_on->tl_core_h : 0;  // vertices of the corner (0: every edge's support comes from the triangle pass)
  info[1] = 0;
  if (on) {
    int e0 = 0;
    HIP_TRY(hipSetDevice(run_on->device));
    HIP_TRY(hipMemcpy(&e0, run_on->d_rp + run_on->tl_skip_from, sizeof(int), hipMemcpyDeviceToHost));
    info[1] = run_on->ne - (long long)e0;  // DAG entries inside it
  }
  info[2] = nb * (nb + 1) / 2;           // pairs of 256-row blocks of the product
  info[3] = on ? (run_on->tl_core_h >> 9) : 0;  // 512-column chunks of a row
  return GM_OK;
}
extern "C" int gm_diamond_support_partial(const gm_graph *sym, const gm_launch *la, uint32_t *d_support, int64_t n_entries, gm_stats *st) {
  if (!d_support) return GM_ERR_INVALID;
  gm_graph *g = const_cast<gm_graph *>(sym), *run_on = nullptr;
  int rc = diamond_run_on(sym, la, &run_on);
  if (rc) return rc;
  const int world = (la && la->world > 1) ? la->world : 1;
  // (at least gm_diamond_support_size; the ranks' slices are cut from THAT size -- size / world entries each -- whatever the buffer holds beyond it)
  if (n_entries < diamond_support_entries(run_on->ne, world)) return GM_ERR_INVALID;
  uint64_t dummy = 0;
  rc = run_pattern(PAT_SUPPORT_PART, run_on, la, 3, (la && la->d_counts) ? nullptr : &dummy, 1, st, -1, 0, d_support);
  if (rc) return rc;
  gm_graph *dag = g->dag_cache;
  dag->ring_alias = (run_on != dag) ? run_on : nullptr;
  g->ring_alias = dag;
  g->ring_extra[0] = g->ring_extra[1] = nullptr;
  return GM_OK;
}
extern "C" int gm_diamond_support_finish(const gm_graph *sym, const gm_launch *la, const uint32_t *d_support, int64_t count, uint64_t *total, gm_stats *st) {
  if (!d_support || count < 0) return GM_ERR_INVALID;
  gm_graph *g = const_cast<gm_graph *>(sym), *run_on = nullptr;
  int rc = diamond_run_on(sym, la, &run_on);
  if (rc) return rc;
  LaunchCtx ctx;
  rc = begin_launch(run_on, la, total, ctx);
  if (rc) return rc;
  rc = start_timer(ctx);
  if (rc) return rc;
  HIP_TRY(launch_sup_pairs(d_support, 0, (long long)count, run_on->d_counters, run_on->cu_count, ctx.stream));
  fill_stats(st, (uint64_t)count, 0, 0, 256);
  rc = end_launch(ctx, FIN_COPY, 0, total, 1, st);
  if (rc) return rc;
  gm_graph *dag = g->dag_cache;
  dag->ring_alias = (run_on != dag) ? run_on : nullptr;
  g->ring_alias = dag;
  g->ring_extra[0] = g->ring_extra[1] = nullptr;
  return GM_OK;
}

static int sgl4_kind(const char *pattern) {  // 1 tailedtriangle, 2 4path, 3 3star, 0 none of them
  if (!pattern) return 0;
  return strcmp(pattern, "tailedtriangle") == 0 ? 1 : strcmp(pattern, "4path") == 0 ? 2 : strcmp(pattern, "3star") == 0 ? 3 : 0;
}

// A rank's share of the four per-edge sums behind tailedtriangle / 4path / 3star (see gm_sgl): plain sums over its tasks, so the shares
// add up; with la->d_counts they are left in that device buffer, ordered on la->stream (raw may be NULL then).
extern "C" int gm_sgl4_partial(const gm_graph *sym, const gm_launch *la, uint64_t raw[4], gm_stats *st) {
  if (!sym || (!raw && !(la && la->d_counts))) return GM_ERR_INVALID;
  if (int rc0 = reject_big(sym)) return rc0;
  const int rc = run_pattern(PAT_MOTIF4E, sym, la, 4, raw, 4, st, FIN_RAW4, 0);
  if (rc) return rc;
  const_cast<gm_graph *>(sym)->ring_alias = nullptr;
  return GM_OK;
}

extern "C" int gm_sgl4_finish(const char *pattern, const uint64_t raw[4], uint64_t *total) {
  const int kind = sgl4_kind(pattern);
  if (!kind || !raw || !total) return GM_ERR_INVALID;
  *total = kind == 1 ? raw[2] / 2 + raw[3] : kind == 2 ? raw[1] + raw[2] + raw[3] : (raw[0] + 2 * raw[2] + 2 * raw[3]) / 6;
  return GM_OK;
}

extern "C" int gm_sgl(const gm_graph *sym, const char *pattern, const gm_launch *la, uint64_t *total, gm_stats *st) {
  if (!pattern) return GM_ERR_INVALID;
  if (strcmp(pattern, "diamond") == 0) {
    // tune[6] & 1024: the LISTING (nested) form of the reference, src/sgl/gpu_kernels/diamond_nested.cuh:4-31 -- materialise
    // S, count_smaller per member -- as a second implementation; the default counts C(|S|,2) per edge (diamond_count.cuh:15-17)
    if (la && (la->tune[6] & 1024)) return run_sgl_nested(SGL_DIAMOND, sym, la, total, st);
    // One GPU: |N(v0) ^ N(v1)| of EVERY edge from one pass over the triangles of the DAG (edge supports, gm_sup.hip), then sum C(t, 2).
    // Several ranks, DAG rows beyond the 2048-entry stage, or one of the A/B switches of the per-edge kernels (tune[6] & 0x10000000: that
    // path on request): one intersection of the two symmetric lists per edge (gm_hrow.hip, gm_chunk.h).
    {
      const int t6 = la ? la->tune[6] : 0;
      const bool big = sym && sym->d_rp64;  // (>= 2^31 entries: only the supports of the oriented copy can run; on one GPU)
      const bool per_edge = ((t6 & (0x10000000 | 0x80000 | 0x100000 | 0x400000 | 0x1000000 | 0x2000000)) || (la && la->world > 1) ||
                             (la && la->tune[5] == 1) || gm_opt("GM_DIAMOND_PER_EDGE") || !sym) && !big;
      if (!per_edge) {
        const int rc = run_diamond_supports(sym, la, total, st);
        if (rc != GM_ERR_UNSUPPORTED) return rc;
        if (big) return reject_big(sym);
      }
    }
    return run_pattern(PAT_DIAMOND, sym, la, 4, total, 1, st);
  }
  // rectangle / house / pentagon run on a copy of the graph renumbered by degree (get_relabeled; tune[6] & 512: on the
  // graph as given). tune[6] & 1024: the wave-per-edge loop nests; & 2048: rectangle as wedges + flat intersections,
  // house without the LDS S-bitmap (A/B, tests).
  const bool is_rect = strcmp(pattern, "rectangle") == 0, is_house = strcmp(pattern, "house") == 0, is_pent = strcmp(pattern, "pentagon") == 0;
  if (is_rect || is_house || is_pent) {
    if (!sym) return GM_ERR_INVALID;
    if (int rc0 = reject_big(sym)) return rc0;
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
  // The other 4-vertex patterns of src/sgl/omp_base.cc:21-31 (round 6) need no enumeration of their own: with tri = |N(u) ^ N(v)|,
  // su = d(u) - tri - 1, sv = d(v) - tri - 1 per undirected edge -- the four per-edge sums of the formula 4-motif (PAT_MOTIF4E,
  // src/motif/cpu_kernels/automine_formula.h:33-41: raw0 = sum su (su - 1) + sv (sv - 1), raw1 = sum su sv, raw2 = sum tri (su + sv),
  // raw3 = sum tri (tri - 1)) --
  //   tailedtriangle (tailedtriangle.h:1-12) = sum_v t(v) (d(v) - 2) = 1/2 sum_e tri (d(u) + d(v) - 4)         = raw2 / 2 + raw3
  //   4path          (4path.h:1-14)          = sum_e (d(u) - 1)(d(v) - 1) - 3 T  (T = 1/3 sum_e tri)            = raw1 + raw2 + raw3
  //   3star          (3star.h:1-13)          = sum_v C(d(v), 3) = 1/6 sum_e [(d(u)-1)(d(u)-2) + (d(v)-1)(d(v)-2)] = (raw0 + 2 raw2 + 2 raw3) / 6
  // (the reference's symmetry breaking makes each of them the plain edge-induced count; checked against sgl_omp_base on the seven golden
  // graphs).  The divisions need the sums of the whole graph: a rank's share goes through gm_sgl4_partial.
  if (sgl4_kind(pattern)) {
    if (la && (la->world > 1 || la->d_counts)) return GM_ERR_UNSUPPORTED;  // several ranks: gm_sgl4_partial + all-reduce + gm_sgl4_finish
    uint64_t raw[4] = {0, 0, 0, 0};
    const int rc = gm_sgl4_partial(sym, la, raw, st);
    return rc ? rc : gm_sgl4_finish(pattern, raw, total);
  }
  if (total) *total = 0;  // "Not implemented", total_num = 0 (src/sgl/omp_base.cc:51-53)
  return GM_ERR_UNSUPPORTED;
}

extern "C" int gm_clique(const gm_graph *dag, int k, const gm_launch *la, uint64_t *total, gm_stats *st) {
  if (k == 3) return run_tc(dag, la, total, 1, st);
  if (k < 3 || k > GM_MAX_CLIQUE_K) {
    if (total) *total = 0;
    return GM_ERR_INVALID;
  }
  if (k == 4 && !(la && (la->tune[6] & (0x40000 | 0x200)))) {
    // the re-hosted first level and the pair counts want a TOPOLOGICAL numbering (upper-triangular matrices: gm_cbuild.hip): a DAG
    // that is not numbered that way runs on its cached renumbered copy (tune[6] & 0x200: on the graph as numbered, like the SgL
    // patterns; & 0x40000: everything in the mining kernel, which does not care)
    gm_graph *self = const_cast<gm_graph *>(dag), *run_on = nullptr;
    int rc = topo_view(dag, la, &run_on);
    if (rc) return rc;
    rc = run_pattern(PAT_CLIQUE4, run_on, la, k, total, 1, st);
    self->ring_alias = (run_on != self) ? run_on : nullptr;
    return rc;
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
  if (int rc0 = reject_big(sym)) return rc0;
  gm_graph *g = const_cast<gm_graph *>(sym);
  gm_launch l2;
  memset(&l2, 0, sizeof l2);
  if (la) l2 = *la;
  uint64_t *d_out = l2.d_counts;
  {  // (the oriented copy, cached on the handle)
    const int rc_dag = ensure_dag_cache(g);
    if (rc_dag) return rc_dag;
  }
  gm_stats s1, s2, s3;
  memset(&s1, 0, sizeof s1); memset(&s2, 0, sizeof s2); memset(&s3, 0, sizeof s3);
  l2.d_counts = d_out;
  int rc = run_pattern(PAT_MOTIF4E, sym, &l2, 4, raw, 4, &s1, FIN_RAW4, 0);
  if (rc) return rc;
  const gm_graph *rect_handle = sym;
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
    rect_handle = rect_on;
  }
  if (rc) return rc;
  l2.d_counts = d_out ? d_out + 5 : nullptr;
  rc = gm_clique(g->dag_cache, 4, &l2, raw ? &raw[5] : nullptr, &s3);  // (on the DAG's topological view)
  if (rc) return rc;
  if (st) {
    *st = s1;
    st->kernel_ms = s1.kernel_ms + s2.kernel_ms + s3.kernel_ms;
  }
  g->ring_alias = nullptr;
  g->ring_extra[0] = (rect_handle != sym) ? rect_handle : nullptr;  // (tune[6] & 512: the rectangle kernel ran on this handle itself)
  g->ring_extra[1] = g->dag_cache;
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
  // Default: the reference's OTHER 3-motif solver (motif_omp_formula / motif_gpu_formula, src/motif/omp_formula.cc:39-46: the triangles
  // of the oriented graph, wedges = sum C(d,2) - 3T) -- the same two counts from 1/4 of the streamed keys, and the only form a graph
  // of 2^31 entries or more fits (its oriented copy has a 32-bit task index).  tune[6] & 0x10000000, or one of the A/B switches of
  // the per-edge kernels: automine_3motif's enumeration (src/motif/cpu_kernels/automine_base.h:2-22), one bounded intersection of the
  // two symmetric lists per edge (gm_hrow.hip, gm_chunk.h).
  const int t6 = la ? la->tune[6] : 0;
  const bool per_edge = ((t6 & (0x10000000 | 0x80000 | 0x100000 | 0x400000 | 0x1000000 | 0x2000000)) || (la && la->tune[5] == 1) ||
                         gm_sweep_env("GM_MOTIF3_PER_EDGE")) && !(sym && sym->d_rp64);
  if (!per_edge) {
    const int rc = gm_motif_formula(sym, k, la, counts, ncounts, st);
    if (rc == GM_OK && st) st->tasks *= 2;  // (the graph's directed entries, as the enumeration reports them: two per task edge of the DAG)
    return rc;
  }
  // The enumeration's two totals do not depend on the vertex numbering, its WORK does: for an edge {lo < hi} only the common neighbours
  // below hi count, and the kernels trim the streamed list to them.  Numbered by DESCENDING degree (the cached copy the SgL wedge forms use),
  // "below hi" = "of higher degree than hi": the trimmed list is what an oriented row would be -- sum_v d+(v)^2 streamed keys instead of
  // sum_e min(d(u), d(v)) over the symmetric lists (R-MAT-24: 150 G).  GM_MOTIF3E_AS_NUMBERED / tune[6] & 512: on the graph as given.
  const gm_graph *run_on = sym;
  if (!(t6 & 512) && !gm_sweep_env("GM_MOTIF3E_AS_NUMBERED")) {
    gm_graph *r = nullptr;
    const int rc = get_relabeled(const_cast<gm_graph *>(sym), 1, &r);
    if (rc == GM_ERR_HIP && g_last_hip_error == (int)hipErrorOutOfMemory) {  // (the copy is an optimisation -- about the size of the graph again: without it the graph runs as numbered)
      (void)hipGetLastError();
      g_last_error.clear();
    } else if (rc) {
      return rc;
    } else {
      run_on = r;
    }
  }
  const int rc = run_pattern(PAT_MOTIF3, run_on, la, 3, counts, ncounts, st);
  if (run_on != sym) const_cast<gm_graph *>(sym)->ring_alias = run_on;
  return rc;
}

// motif_omp_formula / motif_gpu_formula (src/motif/omp_formula.cc:39-46, cpu_kernels/automine_formula.h:2-19):
// enumerate only the triangles, derive the wedges: wedges = sum_v C(d(v),2) - 3*T. Here the triangles come from the
// TC kernel on the oriented graph (built once per handle and cached), so the hub rows of the symmetric graph are
// never intersected. Counts are identical to gm_motif; with world > 1 the sum_v C(d,2) term is contributed by rank 0
// and the per-rank partial wedge count is only meaningful after the all-reduce (mod 2^64 arithmetic).
template <class OffT>
__global__ __launch_bounds__(256) void sum_c2_kernel(int nv, const OffT *__restrict__ rp, unsigned long long *__restrict__ out) {
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
  {  // (the oriented copy, cached on the handle)
    const int rc_dag = ensure_dag_cache(g);
    if (rc_dag) return rc_dag;
  }
  if (!g->sum_c2_valid) {  // sum_v C(d(v),2): one reduction kernel over the offsets
    DevBuf<unsigned long long> acc;
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(acc.alloc(1));
    HIP_TRY(hipMemset(acc.p, 0, 8));
    const dim3 cgrid((unsigned)std::min<long long>(((long long)g->nv + 255) / 256, 2048));
    if (g->d_rp64) hipLaunchKernelGGL((sum_c2_kernel<long long>), cgrid, dim3(256), 0, 0, g->nv, (const long long *)g->d_rp64, acc.p);
    else hipLaunchKernelGGL((sum_c2_kernel<int>), cgrid, dim3(256), 0, 0, g->nv, (const int *)g->d_rp, acc.p);
    unsigned long long s2 = 0;
    HIP_TRY(hipMemcpy(&s2, acc.p, 8, hipMemcpyDeviceToHost));
    g->sum_c2 = s2;
    g->sum_c2_valid = true;
  }
  const int rank = la ? la->rank : 0;
  const int rc = run_tc(g->dag_cache, la, counts, ncounts, st, FIN_MOTIF3_FORMULA, rank == 0 ? g->sum_c2 : 0ull);
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

// tooling (the byte model of bench.py): the blocked gather of the most recent whole-graph 4-clique plan of this handle
extern "C" int gm_clique4_gather_info(const gm_graph *dag, int64_t info[4]) {
  if (!dag || !info) return GM_ERR_INVALID;
  info[0] = info[1] = info[2] = info[3] = 0;
  gm_graph *g = nullptr;
  {  // (the plan belongs to the handle the count runs on: the topologically renumbered copy of a DAG that is not numbered that way)
    const int rc = topo_view(dag, nullptr, &g);
    if (rc) return rc;
  }
  std::lock_guard<std::mutex> lk(g->mu);
  for (auto &pl : g->clique_plans) {
    if (pl.world != 1) continue;
    for (auto &rd : pl.rounds) {
      info[0] += (int64_t)rd.n_gunits;             // (vertex, block) units
      info[1] += (int64_t)rd.gather_table_bytes;   // bytes they read by construction: 16 B record + 4 B per row + their table dwords
      info[2] += (int64_t)rd.n_gitems;             // work items (each loads one block image of <= 64 KB unless it shares it with its predecessor)
    }
    info[3] = g->n_cg_blocks;
    break;
  }
  return GM_OK;
}

// (module warm-up, gm_graph.hip finish_handle: HIP loads the code object of a translation unit when one of its kernels is first launched)
__global__ void gm_touch_launch_kernel() {}
void gm_touch_launch() { hipLaunchKernelGGL(gm_touch_launch_kernel, dim3(1), dim3(1), 0, 0); }
