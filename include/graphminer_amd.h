/*
 * graphminer_amd.h -- C ABI of the MI355X-native subgraph-matching hot path.
 *
 * One shared object (graphminer_amd/libgraphminer_amd.so), plain pointers and sizes,
 * no C++/torch types. It is the drop-in boundary for GraphMiner's link-time solver
 * seam: each reference solver
 *     void TCSolver    (Graph&, uint64_t& total, int n_gpu, int chunk)            src/triangle/main.cc:5
 *     void SglSolver   (Graph&, Pattern&, uint64_t& total, int n_dev, int chunk)  src/sgl/main.cc:7
 *     void CliqueSolver(Graph&, int k, uint64_t& total, int, int)                 src/clique/main.cc:6
 *     void MotifSolver (Graph&, int k, std::vector<uint64_t>&, int, int)          src/motif/main.cc:7
 * becomes a ~10 line shim over gm_tc / gm_sgl / gm_clique / gm_motif (see INTEGRATION.md
 * and graphminer_amd/host/solvers.cc and integration/hip_solvers.cc, which are exactly that shim).
 *
 * Conventions
 *   - every entry point returns a gm_status (0 = ok); nothing calls exit().
 *   - host pointers are borrowed for the duration of the call only.
 *   - device buffers are owned by the opaque gm_graph handle.
 *   - outputs are WRITTEN, not accumulated (the reference multi-GPU solvers '+=' into
 *     a caller-zeroed total, src/triangle/multigpu.cu:84; the shim does that '+=').
 *   - vertex ids int32, CSR offsets int64, counts uint64 (include/common.h:36-40).
 *   - there is NO CPU fallback: without a HIP device every compute call fails with
 *     GM_ERR_NO_DEVICE.
 */
#ifndef GRAPHMINER_AMD_H
#define GRAPHMINER_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gm_status {
  GM_OK = 0,
  GM_ERR_INVALID = 1,      /* bad argument (null pointer, k out of range, ...) */
  GM_ERR_NO_DEVICE = 2,    /* no HIP device / HIP runtime error before launch */
  GM_ERR_HIP = 3,          /* HIP runtime error (see gm_last_error) */
  GM_ERR_TOO_LARGE = 4,    /* outside the 32-bit task index of the mining kernels: a solver asked to walk a graph of >= 2^31 entries
                              itself (such a graph CAN be uploaded, oriented, downloaded and 3-motif-counted), nv >= 2^31-1, or an
                              oriented graph of >= 2^31 entries */
  GM_ERR_UNSUPPORTED = 5,  /* pattern / k not implemented ("Not implemented", src/sgl/omp_base.cc:51) */
  GM_ERR_IO = 6,           /* file could not be opened / short read (custom_alloc.h:38-41) */
  GM_ERR_FORMAT = 7        /* meta.txt violates the loader asserts (src/common/graph.cc:30-34) */
} gm_status;

const char *gm_strerror(int status);
/* thread-local text of the last failing HIP call made through this library */
const char *gm_last_error(void);
int gm_version(void);

/* Exactly what Graph::V()/E()/get_max_degree()/out_rowptr()/out_colidx() expose
 * (include/graph.h:63-64,73,83-84). Host pointers. */
typedef struct gm_csr {
  int32_t nv;
  int64_t ne;
  int32_t max_deg;
  const int64_t *row_ptr; /* nv+1 */
  const int32_t *col_idx; /* ne, rows strictly ascending */
} gm_csr;

/* Device-resident CSR (+ task-chunk tables). Replaces GraphGPU, include/graph_gpu.h:6-211. */
typedef struct gm_graph gm_graph;

int gm_device_count(int *n);

/* GraphGPU::init (include/graph_gpu.h:69-122): H->D copy of row_ptr / col_idx on `device`.
 * ne >= 2^31 (twitter40, friendster: src/triangle/README.md:60-61; eidType is int64, include/common.h:37) gives a BIG handle: it keeps
 * 64-bit offsets and supports gm_graph_orient (the oriented graph must have < 2^31 entries: then gm_tc / gm_clique run on it),
 * gm_motif(k = 3) / gm_motif_formula (the formula solver), gm_graph_meta / _download / _free; every other solver -> GM_ERR_TOO_LARGE. */
int gm_graph_upload(const gm_csr *host, int device, gm_graph **out);
/* Adopt CSR arrays that already live in HBM (e.g. built on the GPU); the arrays are BORROWED
 * and must outlive the handle. d_row_ptr is int64[nv+1], d_col_idx int32[ne]. */
int gm_graph_from_device(int32_t nv, int64_t ne, const int64_t *d_row_ptr, const int32_t *d_col_idx,
                         int device, gm_graph **out);
/* Graph::orientation (src/common/graph.cc:233-279) on the GPU: keep s->d iff
 * deg[d] > deg[s] || (deg[d] == deg[s] && d > s). Returns a new handle. */
int gm_graph_orient(const gm_graph *sym, gm_graph **dag);
/* Graph::sort_neighbors (src/common/graph.cc:138-146), the `adj_sorted = 0` case of tc_* (src/triangle/main.cc:22): sorts
 * every row ascending, in place on the device (one segmented radix sort). Every solver assumes sorted rows, so call it
 * before the first solver / gm_graph_orient result is used; after a solver has run on the handle -> GM_ERR_INVALID. */
int gm_graph_sort_neighbors(gm_graph *g);
/* nv / ne / max_deg of a handle (pointers in *meta are left NULL). */
int gm_graph_meta(const gm_graph *g, gm_csr *meta);
/* D->H copy (row_ptr: nv+1 int64, col_idx: ne int32); either pointer may be NULL. */
int gm_graph_download(const gm_graph *g, int64_t *row_ptr, int32_t *col_idx);
/* The renumbered copy the solvers take for this handle, BORROWED (owned by g, freed with it; do not gm_graph_free it): mode 0 / 1 ids
 * ascending / descending in degree (the SgL kernels), 2 the topological numbering of an oriented graph (TC, k-clique; *view == g when
 * the handle is a DAG that some other rule oriented). A permutation of the same graph with every row ascending; built once on the device
 * (the reference has no counterpart: its kernels walk the graph as numbered, src/common/graph.cc:233-279). For tests and tools. */
int gm_graph_renumbered(gm_graph *g, int mode, gm_graph **view);
void gm_graph_free(gm_graph *g);

/* Scheduler policy for the multi-GPU task split (include/scheduler.h, src/common/scheduler.cc). */
enum {
  GM_PART_ROUND_ROBIN = 0, /* Scheduler::round_robin, scheduler.cc:34-85: the j-th chunk of the dequeue order -> rank
                              j mod world (order = estimated work, heaviest first, when world > 1) */
  GM_PART_RANGE = 1,       /* even contiguous split of the task chunks (EVEN_SPLIT, src/clique/multigpu.cu:42-44) */
  GM_PART_VERTEX = 2       /* vertex-balanced: rank r owns the tasks of vertices [nv*r/world, nv*(r+1)/world)
                              (1-D vertex ranges of src/common/graph_partition.cc:82-132, without the halo copies:
                              the CSR is replicated); solvers only, gm_partition treats it like GM_PART_RANGE */
};

typedef struct gm_launch {
  void *stream;       /* hipStream_t; NULL = the null stream */
  int32_t rank;       /* this process' share of the task chunks: rank in [0, world) */
  int32_t world;      /* 0 or 1 = single GPU */
  int32_t policy;     /* GM_PART_* */
  int32_t chunk;      /* tasks (edges) per scheduling chunk; 0 = library default. Reference default 1024
                         (src/triangle/main.cc:16) */
  uint64_t *d_counts; /* optional DEVICE buffer (>= ncounts uint64). When set the result is left on the
                         device, no host sync is done (use it to feed an RCCL all-reduce). */
  int32_t tune[8];    /* kernel tuning / ablation knobs, all 0 = defaults. Counts never depend on them
                         (tests/test_gpu_parity.py) unless a bit is marked "ablation" (work is skipped, counts wrong).
                         [0] task edges per chunk, [1] chunks per dequeue, [2] xs+1 and [3] ys+1 of the direction rule
                         b*(xb+xs*lg a) <= a*(yb+ys*lg b), [4] workgroups per CU, [5] 1 = never stage adjacency in LDS,
                         [7] bits 0..7 xb*16+yb, bits 8.. price of a pass-Y key against a bitmapped row (0 = as a bisection).
                         [6] bit mask. Alternative implementations (same counts): 0x100 mining kernels ignore the hub bitmaps,
                         0x200 SgL / TC / 4-clique on the graph as numbered (no degree / topological renumbering), 0x400 SgL wave-per-edge loop nests,
                         0x800 rectangle / pentagon as wedges + flat intersections, house flattened over (v0,v1,v3), 0x1000 cut
                         chunks into parts eagerly, 0x2000 swap the two dequeue orders, 0x4000 plain chunk-id order,
                         0x10000000 diamond / 3-motif by one intersection of the two symmetric lists per edge (the reference's loop
                         nests; default: from the triangles of the oriented copy), 0x4000000 TC by the chunked mining kernel (default: the shorter list
                         of every edge against a hashed set, gm_tch.hip),
                         0x20000 rectangle / house with every counter map in global memory,
                         0x800000 hashed sets on their global-memory fallback lookup, 0x40000 4-clique in the mining kernel alone,
                         0x40000000 edge supports (diamond) with one atomic per streamed edge instead of the match masks (gm_sup.hip);
                         0x80000 / 0x100000 / 0x400000 / 0x1000000 / 0x2000000: variants of the per-edge class kernels (gm_launch.hip).
                         Ablation (mining kernels): 0x1 skip clique phase 2, 0x2 skip bit-matrix writes, 0x4 no filter,
                         0x8 / 0x10 / 0x20 filtered-pass stages, 0x40 skip SPLIT chunks, 0x80 only SPLIT chunks,
                         0x400 skip pass X, 0x8000 skip pass Y (house: flattened form without the LDS S-bitmap) (gm_api.hip / gm_mine.hip). */
} gm_launch;

typedef struct gm_stats {
  double kernel_ms;  /* HIP-event time of the mining kernel(s) on the launch stream (0 when d_counts is set
                        and the call did not synchronise) */
  uint64_t tasks;    /* "edges processed" as the reference's TEPS line defines nnz (src/triangle/gpu_base.cu:69) */
  uint64_t chunks;   /* task chunks owned by this rank */
  uint32_t grid;     /* workgroups launched */
  uint32_t block;    /* threads per workgroup */
} gm_stats;

/* The multi-GPU task split as index arithmetic: rank owns chunk ids first + i*step, i in [0,count).
 * Host-only (no device needed). */
int gm_partition(int64_t n_chunks, int32_t rank, int32_t world, int32_t policy, int64_t *first, int64_t *step,
                 int64_t *count);
/* Host-only view of the task-chunk table the solvers build for a CSR: recs receives up to `cap` records
 * {u_begin, u_end, e_begin, e_end} (4 x int32 each); *n_out = number of chunks. for_clique = 1 gives the
 * table of gm_clique (whole rows only). */
int gm_chunk_table(int32_t nv, const int64_t *row_ptr, int32_t chunk, int32_t for_clique, int32_t *recs, int64_t cap,
                   int64_t *n_out);

/* Pre-processing ("setup") accounting. The reference leaves these steps untimed ("Time on generating the DAG",
 * "Time on generating the edgelist", src/common/graph.cc:233-279,297-326); here every handle accumulates the wall-clock
 * milliseconds (host clock around device work + host work, synchronised) it has spent in them, so that a caller can report
 * them beside the kernel time:
 *   orient_ms   gm_graph_orient that produced this handle (recorded on the DAG handle)
 *   table_ms    task-chunk tables: chunk records, cost estimate, parts, dequeue orders (all tables built so far)
 *   bitmap_ms   hub-row bitmaps (symmetric-graph patterns)
 *   relabel_ms  renumbered copies (SgL patterns, clique topological numbering), including their own setup
 *   other_ms    per-pattern tables (idx0, 2-path estimates, per-entry triangle tables, task lists)
 * Cached structures cost nothing on later calls. */
typedef struct gm_setup_times {
  double orient_ms, table_ms, bitmap_ms, relabel_ms, other_ms;
} gm_setup_times;
int gm_graph_setup_times(const gm_graph *g, gm_setup_times *out);

/* HIP-event durations (ms) of the mining kernels of the most recent launches on this handle, oldest
 * first; at most 64 are remembered. The caller must have synchronised the launch stream(s).
 * This is how an asynchronous (d_counts) caller reads the kernel time the reference prints as
 * "runtime [gpu_base]" (src/triangle/gpu_base.cu:53-68). */
int gm_kernel_times(const gm_graph *g, int n, double *ms_out, int *n_out);
/* The part of those durations spent in the launch's hub-corner kernel on the matrix cores (csrc/gm_ctc.hip; 0 for a launch without one):
 * the rest is the streamed kernels' -- the time bench.py prices against the HBM roofline. Same order, same synchronisation rule. */
int gm_corner_times(const gm_graph *g, int n, double *ms_out, int *n_out);

/* ---- solvers ------------------------------------------------------------------------------ */
/* launch may be NULL (single GPU, null stream, defaults). stats may be NULL. */

/* TCSolver: total = sum_{(u,v) in DAG} |N+(u) ^ N+(v)|  (src/triangle/omp_base.cc:15-21,
 * src/triangle/gpu_kernels/bs_warp_edge.cuh:2-18). `dag` must be an ORIENTED graph. */
int gm_tc(const gm_graph *dag, const gm_launch *launch, uint64_t *total, gm_stats *stats);
/* Tooling (tests, the byte model of bench.py): on a DAG whose hubs are its last ids (the topologically renumbered copy of a graph with long
 * rows) the triangles whose smallest member is one of the last H vertices are counted as ONE masked bit-matrix product on the matrix
 * cores (csrc/gm_ctc.hip: sum_{i<j, M_ij} popc(M_i & M_j) over the H x H corner of the adjacency matrix -- the loop of omp_base.cc:15-21 on
 * bit rows) and the streamed kernel takes every other edge.  After a first gm_tc: info[0] = H (0: no such corner on this handle),
 * info[1] = DAG entries inside the corner, info[2] = 64 x 64 blocks of the product, info[3] = vertices of the core bitmap.
 * gm_dev_option("GM_TC_CORE_H", ..) (read when the handle's key stream is built): 0 = off, any other value = that H. */
int gm_tc_core_info(const gm_graph *dag, int64_t info[4]);

/* SglSolver: edge-induced subgraph listing on the SYMMETRIC graph, pattern by NAME
 * (include/pattern.hh:62-78). Implemented: "diamond" (src/sgl/cpu_kernels/diamond.h:1-14,
 * src/sgl/gpu_kernels/diamond_count.cuh:3-21), "rectangle" (rectangle.h:1-11), "house" (house.h:1-16),
 * "pentagon" (pentagon.h:2-17), and -- one rank (several: gm_sgl4_partial), from the per-edge sums of the formula 4-motif, no enumeration of their own --
 * "tailedtriangle" (tailedtriangle.h:1-12), "4path" (4path.h:1-14), "3star" (3star.h:1-13).  Others (the 5- and 6-vertex patterns of
 * src/sgl/omp_base.cc:33-49 beyond house / pentagon) -> GM_ERR_UNSUPPORTED, *total = 0.
 * With world > 1 a rank's house / pentagon value is a partial MODULO 2^64 (a centre's positive and negative terms may be tasks of different
 * ranks): add the ranks' values as uint64 (an all-reduce does), do not compare a single rank's value with anything.
 * rectangle / house keep the counter maps of their heavy centres in LDS (gm_mine.hip rect_lds_kernel / house_lds_kernel; tune[6] & 0x20000:
 * in global memory, round 5's form).
 * diamond = sum over the edges of C(|N(v0) ^ N(v1)|, 2). One GPU: |N(v0) ^ N(v1)| of every edge -- its triangles -- from ONE pass over
 * the triangles of the oriented copy (edge supports, gm_sup.hip; the copy is built and cached on first use; also for a graph of
 * >= 2^31 entries); world > 1, a DAG row beyond 2048 entries, or tune[6] & 0x10000000: one intersection of the two symmetric lists per
 * edge (gm_hrow.hip, gm_chunk.h). Same count. */
int gm_sgl(const gm_graph *sym, const char *pattern, const gm_launch *launch, uint64_t *total, gm_stats *stats);

/* tailedtriangle / 4path / 3star on SEVERAL ranks (the reference's sgl has no multi-GPU binary for them; same contract as
 * gm_motif4_partial): every rank calls gm_sgl4_partial for its share of the four per-edge sums (plain sums over its tasks; with
 * launch->d_counts they are left in that device buffer without synchronising and raw may be NULL), the ranks all-reduce raw[0..3],
 * then gm_sgl4_finish(pattern, raw, &total) applies the pattern's closed form.  gm_sgl itself refuses world > 1 for these three. */
int gm_sgl4_partial(const gm_graph *sym, const gm_launch *launch, uint64_t raw[4], gm_stats *stats);
int gm_sgl4_finish(const char *pattern, const uint64_t raw[4], uint64_t *total);

/* Diamond on SEVERAL ranks with the one-GPU algorithm (one shared pass over the triangles of the oriented copy; the reference has no
 * multi-GPU diamond: src/sgl/multigpu.cu:117 is commented out).  Per step, on every rank:
 *   1. gm_diamond_support_partial: the rank's share (launch->rank / world) of the triangle pass adds its three increments per triangle
 *      into d_support -- the caller's DEVICE buffer of n_entries uint32 (AT LEAST gm_diamond_support_size = S: |E+| of the oriented copy
 *      rounded up so that every rank's slice is equal and 256-byte aligned; fewer -> GM_ERR_INVALID; a larger buffer is accepted and only
 *      its first S entries are used), zeroed by the call, ordered on launch->stream;
 *   2. the caller sums the ranks' arrays with ONE reduce-scatter (ncclReduceScatter, ncclUint32, ncclSum / torch reduce_scatter_tensor):
 *      rank r receives the S / world entries from r * S / world on (S = gm_diamond_support_size, not the buffer's own length);
 *   3. gm_diamond_support_finish: sum C(t, 2) over `count` entries at d_support (the rank's reduced slice) -> total / launch->d_counts;
 *   4. the usual all-reduce of the 64-bit count.
 * Rows of the oriented copy beyond the 2048-entry stage take sup_long_kernel (one wave per edge) inside the same call. */
int gm_diamond_support_size(const gm_graph *sym, int world, int64_t *n_entries);
/* Tooling (bench.py's byte model, tests): after a one-GPU gm_sgl(.., "diamond") on this handle -- info[0] = 64-bit words of the match-mask
 * arena (the in-edge tasks of the triangle pass report their streamed edges as bit masks of their tails, gm_sup.hip; 0 = no masks on this
 * graph), info[1] = streamed-edge increments the most recent launch issued as global atomics, info[2] = rows with masks of several words,
 * info[3] = the shortest tail that gets a mask.  Synchronises the device. */
int gm_diamond_support_info(const gm_graph *sym, int64_t info[4]);
/* Tooling: the supports of the edges inside the hub corner (the last H vertices of the renumbered oriented copy) are one bit-matrix product
 * on the matrix cores, t(i, j) = (A A)_ij over the symmetric corner A (csrc/gm_ctc.hip), and the triangle pass leaves the corner's rows
 * out: info[0] = H (0: none), info[1] = DAG entries inside the corner, info[2] = pairs of 256-row blocks, info[3] = 512-column chunks per
 * row.  gm_dev_option("GM_SUP_CORE_H", ..) (read when the handle's task lists are built): 0 = off, a multiple of 512 = that H. */
int gm_sup_core_info(const gm_graph *sym, int64_t info[4]);
int gm_diamond_support_partial(const gm_graph *sym, const gm_launch *launch, uint32_t *d_support, int64_t n_entries, gm_stats *stats);
int gm_diamond_support_finish(const gm_graph *sym, const gm_launch *launch, const uint32_t *d_support, int64_t count, uint64_t *total,
                              gm_stats *stats);

#define GM_MAX_CLIQUE_K 12
/* CliqueSolver on the DAG, 3 <= k <= GM_MAX_CLIQUE_K (src/clique/cpu_kernels/automine_omp.h:67-83,138-157;
 * src/clique/gpu_kernels/clique4_warp_edge.cuh:3-31 ... clique8; k = 9..12: the same levels once more, as the reference's generic
 * clique_omp_recursive / edge_warp_iterative.cuh:2-75 count them -- its gpu_base.cu:59-71 stops at 8), any out-degree. k = 4: the first DFS level is re-hosted (every
 * edge at the endpoint with the longer out-list, gm_cbuild.hip), rows of up to 2048 entries keep their adjacency bit-matrix in an
 * arena in HBM, longer rows one per workgroup. k >= 5: the deeper levels run on induced sub-matrices; rows of up to 4096 entries sweep
 * them with two words per lane, longer rows out of the workgroup's global scratch (slow, exact; the degree-ordered DAG of com-Orkut has
 * rows of at most 535 entries, of a scale-24 R-MAT graph 1744). The workgroups' scratch slots must fit the device memory, else
 * GM_ERR_TOO_LARGE -- a count is refused, never wrong. TC (k = 3) and k = 4 run on a topologically renumbered copy of a DAG whose
 * rows are long (cached on the handle; tune[6] & 0x200: as numbered). */
int gm_clique(const gm_graph *dag, int k, const gm_launch *launch, uint64_t *total, gm_stats *stats);

/* MotifSolver on the SYMMETRIC graph. k = 3: counts[0] = wedges, counts[1] = triangles
 * (the CPU order, src/motif/cpu_kernels/automine_base.h:13,18; NOT the swapped order of
 * src/motif/gpu_kernels/motif3_edge_warp.cuh:19-22). ncounts must be
 * num_possible_patterns[k] (include/pattern.hh:4-15): 2 for k = 3, 6 for k = 4 (see gm_motif4_partial).
 * k = 3 takes the reference's formula solver by default (gm_motif_formula below: the triangles of the oriented copy, wedges derived);
 * tune[6] & 0x10000000: automine_3motif's enumeration, one bounded intersection of the two symmetric lists per edge. Same counts;
 * stats->tasks = the graph's directed entries either way.
 * With world > 1 a rank's wedge value is a partial modulo 2^64 (formula: rank 0 contributes sum C(d,2); enumeration: one intersection
 * per undirected edge serves both directed edges, which may belong to different ranks); the uint64 sum over ranks is the exact count. */
int gm_motif(const gm_graph *sym, int k, const gm_launch *launch, uint64_t *counts, int ncounts, gm_stats *stats);

/* 4-motif in the reference's formula form (src/motif/cpu_kernels/automine_formula.h:21-56, host fix-up
 * src/motif/omp_formula.cc:41-45). gm_motif(k = 4, ncounts = 6) returns [3-star, 4-path, tailed-triangle, 4-cycle,
 * diamond, 4-clique] (vertex-induced, the order of src/motif/README.md:50-60) on one GPU. Multi-GPU: every rank calls
 * gm_motif4_partial (raw[6] are plain sums over its tasks), the ranks all-reduce raw, then gm_motif4_finish.
 * Asynchronous form: with launch->d_counts set, gm_motif4_partial leaves raw[0..5] in that device buffer without
 * synchronising (raw may then be NULL) -- all-reduce it in place, copy it back, call gm_motif4_finish. gm_motif(k = 4)
 * accepts counts == NULL with d_counts set like every other solver (the fix-up then runs on the device). */
int gm_motif4_partial(const gm_graph *sym, const gm_launch *launch, uint64_t raw[6], gm_stats *stats);
int gm_motif4_finish(const uint64_t raw[6], uint64_t counts[6]);

/* motif_omp_formula / motif_gpu_formula (src/motif/omp_formula.cc:39-46, src/motif/gpu_formula.cu:86-92): same
 * counts as gm_motif(k = 3), obtained by enumerating only the triangles (TC kernel on the oriented graph, built once
 * per handle) and deriving wedges = sum_v C(d(v),2) - 3T. */
int gm_motif_formula(const gm_graph *sym, int k, const gm_launch *launch, uint64_t *counts, int ncounts, gm_stats *stats);

/* ---- set-operation primitives (batch form) -------------------------------------------------- */
/* The wave64 equivalents of include/set_intersect.cuh / set_difference.cuh, exposed so the
 * per-primitive parity tests can drive them directly. One wave per pair of ascending int32 lists.
 * All pointers are DEVICE pointers. */
enum {
  GM_OP_INTERSECT_NUM = 0,       /* |A ^ B|                                  set_intersect.cuh:352  */
  GM_OP_INTERSECT_NUM_UPPER = 1, /* |{x in A ^ B : x < upper_i}|              set_intersect.cuh:392  */
  GM_OP_INTERSECT_SET = 2,       /* A ^ B -> out (ascending)                  set_intersect.cuh:73   */
  GM_OP_DIFFERENCE_NUM = 3,      /* |{x in A : x not in B}|                   set_difference.cuh:20  */
  GM_OP_DIFFERENCE_NUM_UPPER = 4,/* |{x in A : x < upper_i, x not in B}|      set_difference.cuh:62  */
  GM_OP_DIFFERENCE_SET = 5,      /* A \ B -> out (ascending)                  set_difference.cuh:112 */
  GM_OP_INTERSECT_SET_UPPER = 6, /* {x in A ^ B : x < upper_i} -> out         set_intersect.cuh:152  */
  GM_OP_DIFFERENCE_SET_UPPER = 7,/* {x in A \ B : x < upper_i} -> out         set_difference.cuh:171 */
  GM_OP_COUNT_SMALLER = 8        /* |{x in A : x < upper_i}| (B unused)       operations.cuh:61-105  */
};
/* Pair i is A_i = values[a_begin[i] .. a_end[i]), B_i = values[b_begin[i] .. b_end[i]).
 * out_num: npairs uint32 (count, or size of the materialised set). For *_SET ops the result of pair
 * i is written at out_values[a_begin[i] ..) (capacity |A_i|; may be NULL for count-only ops).
 * d_upper may be NULL for ops without a bound. d_skip (may be NULL) is the CPU-oracle "other.vid"
 * exclusion of the difference ops (src/common/VertexSet.cc:29,37): elements equal to skip[i] are dropped. */
int gm_setop_batch(int op, int64_t npairs, const int32_t *d_values, const int64_t *d_a_begin, const int64_t *d_a_end,
                   const int64_t *d_b_begin, const int64_t *d_b_end, const int32_t *d_upper, const int32_t *d_skip,
                   uint32_t *d_out_num, int32_t *d_out_values, void *stream);

/* ---- tooling ------------------------------------------------------------------------------- */
/* Deterministic R-MAT edge generator: edge i of 2^scale*edge_factor,
 * quadrant probabilities (0.57, 0.19, 0.19, 0.05), SplitMix64 counter hash of (seed, i, level).
 * Writes 2 keys per edge (src<<32|dst and dst<<32|src) to DEVICE buffer d_keys[2*n_edges];
 * self-loops are written as the sentinel 0xFFFFFFFFFFFFFFFF. */
int gm_rmat_keys(int scale, int64_t n_edges, uint64_t seed, uint64_t *d_keys, void *stream);

/* Tooling: level-2 part of the ALGORITHMIC bytes of one 4-clique launch on `dag`:
 * sum_e [8|S1| + sum_{v2 in S1}(4(|S1| + d+(v2)) + 16)]; the level-1 part is the TC formula. Runs one statistics
 * kernel (per-edge |S1| and the out-degrees of the matched vertices). */
int gm_clique4_level2_bytes(const gm_graph *dag, uint64_t *bytes);

/* Tooling (bench.py's byte model): after a whole-graph gm_clique(dag, 4) -- the blocked gather of the wide vertices' core rows
 * (csrc/gm_cgather.hip): info[0] = (vertex, block of core rows) units, info[1] = the bytes they read by construction (a 16-byte record, 4 B per
 * row, the 16-bit column table from the unit's first row on), info[2] = work items, info[3] = blocks of the core.  All 0: the row-major gather. */
int gm_clique4_gather_info(const gm_graph *dag, int64_t info[4]);

/* PMC calibration (tooling): one pass of dword-per-lane coalesced loads over d_buf[0..n), sum -> *d_out.
 * Exactly 4n bytes are read once; run under `rocprofv3 --pmc FETCH_SIZE` to get the counter scale for the
 * access width the mining kernels use (MI355X_MICROARCH.md: FETCH_SIZE is calibrated only for 16 B/lane). */
int gm_calib_stream(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream);

/* Measured stream ceiling (tooling): one pass of 16-byte-per-lane loads over d_buf[0..n) (n int32 words, 16-byte aligned),
 * xor-sum -> *d_out. Time it with events: bytes / t is the read bandwidth a plain stream reaches on this part (the MEASURED
 * ceiling bench.py prints beside the 8 TB/s spec). */
int gm_stream_ceiling(const int32_t *d_buf, int64_t n, uint64_t *d_out, void *stream);

/* Issue-rate calibration (tooling): `waves_per_simd` (1..8) waves on every SIMD of every CU each execute iters x 64 instructions of
 * one kind (0 v_add_u32, 1 v_mul_lo_u32, 2 v_mul_u32_u24, 3 v_cmp -> SGPR pair, 4 v_add_u32 DPP row_shr, 5 s_add_u32, 6 ds_read_b128,
 * 7 ds_read_b32, 8 ds_read_b32 with 32-way bank conflicts, 9 v_add + s_add interleaved, 10 v_readlane, 11 v_mbcnt, 12 v_bcnt,
 * 13 v_cmp SDWA, 14 ds_write_b32, 15 v_cndmask) between two s_memtime reads. Outputs: shader cycles per wave-instruction as one wave
 * sees it (median over the waves) and wave-instructions per cycle and SIMD. The basis of every "unit X is N % busy" statement in
 * DESIGN.md (scripts/issue_calibration.py -> profiles/r03/issue_calibration.txt). Uses the null stream of the current device. */
int gm_issue_calib(int kind, int waves_per_simd, int iters, double *cycles_per_wave_inst, double *inst_per_cycle_simd, double *ms,
                   double *residency /* measured: waves of the launch in flight per SIMD at the same time (mean over SIMDs of the maximum) */);

/* Tooling: a kernel constant by name ("tct_stage_max", "topo_min_mean_row", "motif_trim_min_list", "cb_min_deg", "cb_max_deg",
 * "long_list", "stage_cap", "default_chunk", "mma_words_small", "mma_words_big", "wide_max_deg", "bit_words") -- what the byte model
 * of bench.py needs to know about the kernels, read from the headers they are compiled with. GM_ERR_INVALID for an unknown name. */
int gm_constant(const char *name, int64_t *value);

/* Developer / test options (tooling).  The library reads NO algorithm switch from the environment; the named switches that tests and A/B
 * runs need -- forcing a path on a small graph, the documented fallbacks (csrc/gm_mine.h lists them) -- are set here, process-wide:
 * value = NULL removes `name`, name = NULL removes every option.  Options that shape a handle's cached tables are read when those tables
 * are built (use a fresh handle after changing one).  gm_dev_option_get: the current value or NULL.  Nothing in a production caller
 * needs either. */
int gm_dev_option(const char *name, const char *value);
const char *gm_dev_option_get(const char *name);

/* wave-primitive self test (DPP scans, ballot rank, LDS search); returns GM_OK when the device
 * results equal the host expectation. *n_fail receives the number of mismatching lanes. */
int gm_selftest(int device, int *n_fail);

#ifdef __cplusplus
}
#endif
#endif
