/*
 * gm_oracle.h -- CPU ORACLE for the subgraph-matching hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C (OpenMP) restatement of the
 * reference's CPU algorithms (chenxuhao/GraphMiner @ 2024_10_08). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The
 * product path (graphminer_amd/, include/graphminer_amd.h) never links, loads or
 * calls anything declared here.
 *
 * PARITY PINNED: every solver below is checked count-for-count against
 *   (i)  the README known-answer tables of the reference (tests/golden/golden.json), and
 *   (ii) the reference's own binaries compiled from /root/reference by
 *        oracle/ref/Makefile into oracle/_ref/ (tests/test_oracle_vs_ref.py).
 *
 * Types follow include/common.h:36-40 of the reference:
 *   vertex id int32, edge offset int64, counts uint64.
 */
#ifndef GM_ORACLE_H
#define GM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t gmo_vid;
typedef int64_t gmo_eid;

/* A CSR graph view (borrowed pointers unless produced by gmo_load/gmo_orient). */
typedef struct gmo_graph {
  gmo_vid nv;
  gmo_eid ne;
  gmo_vid max_degree;
  gmo_eid *row_ptr; /* nv+1 */
  gmo_vid *col_idx; /* ne, each row strictly ascending */
} gmo_graph;

/* ---- L0: graph IO / orientation / edge list ------------------------------ */
/* Graph::Graph, src/common/graph.cc:4-42 (meta.txt + vertex.bin + edge.bin). 0 on success. */
int gmo_load(const char *prefix, gmo_graph *out);
void gmo_free(gmo_graph *g);
/* Graph::orientation, src/common/graph.cc:233-279. Allocates out->row_ptr/col_idx. */
int gmo_orient(const gmo_graph *sym, gmo_graph *out);
/* Graph::init_edgelist(sym_break, ascend=false), src/common/graph.cc:297-326.
 * src/dst must hold ne (or ne/2 when sym_break) entries. Returns the number written. */
gmo_eid gmo_edgelist(const gmo_graph *g, int sym_break, gmo_vid *src, gmo_vid *dst);

/* ---- L1: set operations on raw ascending int32 slices -------------------- */
/* Appendix B of SURVEY.md; bvid = B.vid (the vertex whose row B is), -1 for scratch sets. */
uint32_t gmo_intersect_num(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b);
uint32_t gmo_intersect_num_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper);
gmo_vid gmo_intersect_set(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid *out);
gmo_vid gmo_intersect_set_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper, gmo_vid *out);
uint32_t gmo_intersect_num_except2(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid x, gmo_vid y);
uint32_t gmo_intersect_num_upper_except(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper, gmo_vid x);
gmo_vid gmo_difference_set(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid *out);
gmo_vid gmo_difference_set_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid upper, gmo_vid *out);
uint32_t gmo_difference_num_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid upper);
gmo_vid gmo_bounded(const gmo_vid *A, gmo_vid a, gmo_vid up);

/* ---- L3: solvers (vertex-parallel, schedule(dynamic,1), merge set ops) ---- */
/* All take the graph the reference's main would hand the solver:
 *   tc / clique : the ORIENTED (DAG) graph   (src/triangle/main.cc:14, src/clique/main.cc:16)
 *   diamond / sgl patterns / motif : the SYMMETRIC graph (src/sgl/main.cc:16, src/motif/main.cc:15)
 * Range variants [v_begin, v_end) exist so bench.py can time a bounded sample. */
uint64_t gmo_tc(const gmo_graph *dag);
uint64_t gmo_tc_range(const gmo_graph *dag, gmo_vid v_begin, gmo_vid v_end);
/* vertices u = offset (mod stride): bounded sample for bench.py's cpu_baseline */
uint64_t gmo_tc_sample(const gmo_graph *dag, gmo_vid stride, gmo_vid offset, uint64_t *tasks);
uint64_t gmo_diamond(const gmo_graph *sym);
uint64_t gmo_diamond_range(const gmo_graph *sym, gmo_vid v_begin, gmo_vid v_end);
/* strided samples like gmo_tc_sample (vertices v0 = offset mod stride); *tasks = the tasks of the sampled vertices as the
 * reference counts them (diamond: edges v1 < v0; clique: DAG edges; motif: directed edges) */
uint64_t gmo_diamond_sample(const gmo_graph *sym, gmo_vid stride, gmo_vid offset, uint64_t *tasks);
uint64_t gmo_clique_sample(const gmo_graph *dag, int k, gmo_vid stride, gmo_vid offset, uint64_t *tasks);
void gmo_motif3_sample(const gmo_graph *sym, gmo_vid stride, gmo_vid offset, uint64_t out[2], uint64_t *tasks);
uint64_t gmo_rectangle(const gmo_graph *sym);
uint64_t gmo_house(const gmo_graph *sym);
uint64_t gmo_3star(const gmo_graph *sym);          /* src/sgl/cpu_kernels/3star.h */
uint64_t gmo_4path(const gmo_graph *sym);          /* src/sgl/cpu_kernels/4path.h */
uint64_t gmo_tailedtriangle(const gmo_graph *sym); /* src/sgl/cpu_kernels/tailedtriangle.h */
uint64_t gmo_pentagon(const gmo_graph *sym);
/* k in {3,4,5}: the reference's automine loops; k in 6..8: same DFS written recursively. */
uint64_t gmo_clique(const gmo_graph *dag, int k);
uint64_t gmo_clique_range(const gmo_graph *dag, int k, gmo_vid v_begin, gmo_vid v_end);
void gmo_motif3(const gmo_graph *sym, uint64_t out[2]);
void gmo_motif3_range(const gmo_graph *sym, gmo_vid v_begin, gmo_vid v_end, uint64_t out[2]);
void gmo_motif4(const gmo_graph *sym, uint64_t out[6]);

/* ---- ALGORITHMIC bytes (exact, one pass) ----------- */
uint64_t gmo_alg_bytes_tc(const gmo_graph *dag);
uint64_t gmo_alg_bytes_diamond(const gmo_graph *sym);
uint64_t gmo_alg_bytes_clique4(const gmo_graph *dag);
uint64_t gmo_alg_bytes_motif3(const gmo_graph *sym);

int gmo_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
