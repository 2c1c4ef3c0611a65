/*
 * gm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see gm_oracle.h).
 *
 * Plain-C/OpenMP restatement of the reference's CPU path
 * (chenxuhao/GraphMiner @ 2024_10_08). Every function cites the reference
 * file:line it follows. Parity is PINNED: tests/test_oracle_golden.py checks the
 * README known answers and tests/test_oracle_vs_ref.py checks the reference's own
 * binaries (oracle/_ref/, built by oracle/ref/Makefile) on citeseer, cora and
 * seeded RMAT graphs.
 */
#include "gm_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* L0: IO                                                                     */
/* ------------------------------------------------------------------------- */

static int read_exact(const char *path, void *dst, size_t bytes) {
  /* read_file, include/custom_alloc.h:34-44 (open failure is fatal there; an error here) */
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  size_t got = fread(dst, 1, bytes, f);
  fclose(f);
  return got == bytes ? 0 : -2;
}

int gmo_load(const char *prefix, gmo_graph *out) {
  /* Graph::Graph, src/common/graph.cc:19-42: meta.txt = nv ne vid_size eid_size
   * vlabel_size elabel_size max_degree feat_len n_vlabels n_elabels */
  char path[4096];
  memset(out, 0, sizeof(*out));
  snprintf(path, sizeof path, "%s.meta.txt", prefix);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  long long nv = 0, ne = 0;
  int vs = 0, es = 0, vls = 0, els = 0, md = 0, fl = 0, nvc = 0, nec = 0;
  int n = fscanf(f, "%lld %lld %d %d %d %d %d %d %d %d", &nv, &ne, &vs, &es, &vls, &els, &md, &fl, &nvc, &nec);
  fclose(f);
  if (n < 7) return -2;
  if (vs != 4 || es != 8) return -3;          /* graph.cc:30-31 asserts */
  if (!(md > 0 && md < nv)) return -4;        /* graph.cc:34 assert */
  out->nv = (gmo_vid)nv;
  out->ne = (gmo_eid)ne;
  out->max_degree = md;
  out->row_ptr = (gmo_eid *)malloc(sizeof(gmo_eid) * (size_t)(nv + 1));
  out->col_idx = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)(ne > 0 ? ne : 1));
  snprintf(path, sizeof path, "%s.vertex.bin", prefix);
  if (read_exact(path, out->row_ptr, sizeof(gmo_eid) * (size_t)(nv + 1))) { gmo_free(out); return -5; }
  snprintf(path, sizeof path, "%s.edge.bin", prefix);
  if (read_exact(path, out->col_idx, sizeof(gmo_vid) * (size_t)ne)) { gmo_free(out); return -6; }
  return 0;
}

void gmo_free(gmo_graph *g) {
  free(g->row_ptr);
  free(g->col_idx);
  g->row_ptr = NULL;
  g->col_idx = NULL;
}

static inline gmo_vid deg_of(const gmo_graph *g, gmo_vid v) { return (gmo_vid)(g->row_ptr[v + 1] - g->row_ptr[v]); }
static inline const gmo_vid *row_of(const gmo_graph *g, gmo_vid v) { return g->col_idx + g->row_ptr[v]; }

int gmo_orient(const gmo_graph *sym, gmo_graph *out) {
  /* Graph::orientation, src/common/graph.cc:233-279.
   * keep (s->d) iff deg[d] > deg[s] || (deg[d] == deg[s] && d > s), degrees on the symmetric input. */
  gmo_vid nv = sym->nv;
  gmo_vid *newdeg = (gmo_vid *)calloc((size_t)nv + 1, sizeof(gmo_vid));
#pragma omp parallel for schedule(static)
  for (gmo_vid s = 0; s < nv; s++) {
    gmo_vid ds = deg_of(sym, s), c = 0;
    const gmo_vid *r = row_of(sym, s);
    for (gmo_vid i = 0; i < ds; i++) {
      gmo_vid d = r[i], dd = deg_of(sym, d);
      if (dd > ds || (dd == ds && d > s)) c++;
    }
    newdeg[s] = c;
  }
  out->nv = nv;
  out->row_ptr = (gmo_eid *)malloc(sizeof(gmo_eid) * ((size_t)nv + 1));
  gmo_eid acc = 0;
  gmo_vid md = 0;
  for (gmo_vid v = 0; v < nv; v++) { /* parallel_prefix_sum, include/scan.h:5-35 (serial here) */
    out->row_ptr[v] = acc;
    acc += newdeg[v];
    if (newdeg[v] > md) md = newdeg[v];
  }
  out->row_ptr[nv] = acc;
  out->ne = acc;
  out->max_degree = md; /* graph.cc:252 */
  out->col_idx = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)(acc > 0 ? acc : 1));
#pragma omp parallel for schedule(static)
  for (gmo_vid s = 0; s < nv; s++) {
    gmo_vid ds = deg_of(sym, s);
    const gmo_vid *r = row_of(sym, s);
    gmo_vid *w = out->col_idx + out->row_ptr[s];
    for (gmo_vid i = 0; i < ds; i++) {
      gmo_vid d = r[i], dd = deg_of(sym, d);
      if (dd > ds || (dd == ds && d > s)) *w++ = d;
    }
  }
  free(newdeg);
  return 0;
}

gmo_eid gmo_edgelist(const gmo_graph *g, int sym_break, gmo_vid *src, gmo_vid *dst) {
  /* Graph::init_edgelist(sym_break, ascend=false), src/common/graph.cc:308-320 */
  gmo_eid i = 0;
  for (gmo_vid v = 0; v < g->nv; v++) {
    gmo_vid d = deg_of(g, v);
    const gmo_vid *r = row_of(g, v);
    for (gmo_vid k = 0; k < d; k++) {
      gmo_vid u = r[k];
      if (u == v) continue;            /* :310 no selfloops */
      if (sym_break && v < u) break;   /* :314 */
      src[i] = v;
      dst[i] = u;
      i++;
    }
  }
  return i;
}

/* ------------------------------------------------------------------------- */
/* L1: set operations (two-pointer merge)                                     */
/* ------------------------------------------------------------------------- */

uint32_t gmo_intersect_num(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b) {
  /* VertexSet::get_intersect_num, include/VertexSet.h:65-76 */
  uint32_t num = 0;
  gmo_vid l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y) num++;
  }
  return num;
}

uint32_t gmo_intersect_num_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper) {
  /* VertexSet::intersect_ns, include/VertexSet.h:110-122 */
  uint32_t num = 0;
  gmo_vid l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x >= upper) break;
    if (y >= upper) break;
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y) num++;
  }
  return num;
}

gmo_vid gmo_intersect_set(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid *out) {
  /* VertexSet::operator&, include/VertexSet.h:53-64 */
  gmo_vid n = 0, l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y) out[n++] = x;
  }
  return n;
}

gmo_vid gmo_intersect_set_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper, gmo_vid *out) {
  /* VertexSet::intersect(other, upper), include/VertexSet.h:95-108 */
  gmo_vid n = 0, l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x >= upper) break;
    if (y >= upper) break;
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y) out[n++] = x;
  }
  return n;
}

uint32_t gmo_intersect_num_except2(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid ex, gmo_vid ey) {
  /* VertexSet::intersect_ns_except(other, ancestorA, ancestorB), include/VertexSet.h:178-188 */
  uint32_t num = 0;
  gmo_vid l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y && x != ex && x != ey) num++;
  }
  return num;
}

uint32_t gmo_intersect_num_upper_except(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid upper, gmo_vid ex) {
  /* VertexSet::intersect_ns_bound_except, include/VertexSet.h:152-164 */
  uint32_t num = 0;
  gmo_vid l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x >= upper) break;
    if (y >= upper) break;
    if (x <= y) l++;
    if (y <= x) r++;
    if (x == y && x != ex) num++;
  }
  return num;
}

gmo_vid gmo_difference_set(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid *out) {
  /* VertexSet::difference_buf(outBuf, other), src/common/VertexSet.cc:21-43
   * NB: also drops other.vid (":29,:37"). out may alias A. */
  gmo_vid n = 0, l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x <= y) l++;
    if (y <= x) r++;
    if (x < y && x != bvid) out[n++] = x;
  }
  while (l < a) {
    gmo_vid x = A[l];
    l++;
    if (x != bvid) out[n++] = x;
  }
  return n;
}

gmo_vid gmo_difference_set_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid upper, gmo_vid *out) {
  /* VertexSet::difference_buf(outBuf, other, upper), src/common/VertexSet.cc:46-67 */
  gmo_vid n = 0, l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x >= upper) break;
    if (y >= upper) break;
    if (x <= y) l++;
    if (y <= x) r++;
    if (x < y && x != bvid) out[n++] = x;
  }
  while (l < a) {
    gmo_vid x = A[l];
    if (x >= upper) break;
    l++;
    if (x != bvid) out[n++] = x;
  }
  return n;
}

uint32_t gmo_difference_num_upper(const gmo_vid *A, gmo_vid a, const gmo_vid *B, gmo_vid b, gmo_vid bvid, gmo_vid upper) {
  /* VertexSet::difference_ns, src/common/VertexSet.cc:69-89 */
  uint32_t n = 0;
  gmo_vid l = 0, r = 0;
  while (l < a && r < b) {
    gmo_vid x = A[l], y = B[r];
    if (x >= upper) break;
    if (y >= upper) break;
    if (x <= y) l++;
    if (y <= x) r++;
    if (x < y && x != bvid) n++;
  }
  while (l < a) {
    gmo_vid x = A[l];
    if (x >= upper) break;
    l++;
    if (x != bvid) n++;
  }
  return n;
}

gmo_vid gmo_bounded(const gmo_vid *A, gmo_vid a, gmo_vid up) {
  /* VertexSet::bounded, include/VertexSet.h:240-255 (binary search above 64 entries, else linear) */
  if (a > 64) {
    gmo_vid l = -1, r = a;
    while (r - l > 1) {
      gmo_vid t = (l + r) / 2;
      if (A[t] < up) l = t; else r = t;
    }
    return l + 1;
  }
  gmo_vid l = 0;
  while (l < a && A[l] < up) ++l;
  return l;
}

/* ------------------------------------------------------------------------- */
/* L3: solvers                                                                */
/* ------------------------------------------------------------------------- */

int gmo_num_threads(void) {
  int n = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
#pragma omp single
    n = omp_get_num_threads();
  }
#endif
  return n;
}

uint64_t gmo_tc_range(const gmo_graph *g, gmo_vid vb, gmo_vid ve) {
  /* TCSolver, src/triangle/omp_base.cc:14-21 */
  uint64_t counter = 0;
#pragma omp parallel for reduction(+ : counter) schedule(dynamic, 1)
  for (gmo_vid u = vb; u < ve; u++) {
    const gmo_vid *yu = row_of(g, u);
    gmo_vid du = deg_of(g, u);
    for (gmo_vid i = 0; i < du; i++) {
      gmo_vid v = yu[i];
      counter += (uint64_t)gmo_intersect_num(yu, du, row_of(g, v), deg_of(g, v));
    }
  }
  return counter;
}
uint64_t gmo_tc(const gmo_graph *g) { return gmo_tc_range(g, 0, g->nv); }

uint64_t gmo_tc_sample(const gmo_graph *g, gmo_vid stride, gmo_vid offset, uint64_t *tasks) {
  /* same loop as gmo_tc over the vertices u = offset (mod stride): a BOUNDED, degree-representative
   * sample of the workload for bench.py's cpu_baseline (R-MAT ids correlate with degree, so a
   * contiguous range would not be representative). *tasks = DAG edges owned by the sampled vertices. */
  uint64_t counter = 0, t = 0;
  if (stride < 1) stride = 1;
  gmo_vid n = (g->nv - offset + stride - 1) / stride;
#pragma omp parallel for reduction(+ : counter, t) schedule(dynamic, 1)
  for (gmo_vid i = 0; i < n; i++) {
    gmo_vid u = offset + i * stride;
    const gmo_vid *yu = row_of(g, u);
    gmo_vid du = deg_of(g, u);
    t += (uint64_t)du;
    for (gmo_vid k = 0; k < du; k++) {
      gmo_vid v = yu[k];
      counter += (uint64_t)gmo_intersect_num(yu, du, row_of(g, v), deg_of(g, v));
    }
  }
  if (tasks) *tasks = t;
  return counter;
}

static uint64_t diamond_vertex(const gmo_graph *g, gmo_vid v0, gmo_vid *buf, uint64_t *tasks) {
  /* one iteration of the v0 loop of src/sgl/cpu_kernels/diamond.h:1-14 (pair enumeration kept literal: it IS the CPU
   * baseline's cost); *tasks += symmetry-broken edges (v0, v1 < v0) */
  uint64_t counter = 0;
  const gmo_vid *y0 = row_of(g, v0);
  gmo_vid d0 = deg_of(g, v0);
  for (gmo_vid i = 0; i < d0; i++) {
    gmo_vid v1 = y0[i];
    if (v1 >= v0) break;
    *tasks += 1;
    gmo_vid n = gmo_intersect_set(y0, d0, row_of(g, v1), deg_of(g, v1), buf);
    for (gmo_vid p = 0; p < n; p++) {
      gmo_vid v2 = buf[p];
      for (gmo_vid q = 0; q < n; q++) {
        if (buf[q] >= v2) break;
        counter += 1;
      }
    }
  }
  return counter;
}

/* vertices v0 = begin + i * stride < end: stride 1 = the contiguous range, stride > 1 = a degree-representative sample */
static uint64_t diamond_strided(const gmo_graph *g, gmo_vid vb, gmo_vid ve, gmo_vid stride, uint64_t *tasks) {
  uint64_t counter = 0, t = 0;
  if (stride < 1) stride = 1;
  gmo_vid n = ve > vb ? (ve - vb + stride - 1) / stride : 0;
#pragma omp parallel reduction(+ : counter, t)
  {
    gmo_vid *buf = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)(g->max_degree > 0 ? g->max_degree : 1));
#pragma omp for schedule(dynamic, 1)
    for (gmo_vid i = 0; i < n; i++) counter += diamond_vertex(g, vb + i * stride, buf, &t);
    free(buf);
  }
  if (tasks) *tasks = t;
  return counter;
}
uint64_t gmo_diamond_range(const gmo_graph *g, gmo_vid vb, gmo_vid ve) { return diamond_strided(g, vb, ve, 1, NULL); }
uint64_t gmo_diamond_sample(const gmo_graph *g, gmo_vid stride, gmo_vid offset, uint64_t *tasks) {
  return diamond_strided(g, offset, g->nv, stride, tasks);
}
uint64_t gmo_diamond(const gmo_graph *g) { return gmo_diamond_range(g, 0, g->nv); }

uint64_t gmo_rectangle(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/rectangle.h:1-11 */
  uint64_t counter = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : counter)
  for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
    const gmo_vid *y0 = row_of(g, v0);
    gmo_vid d0 = deg_of(g, v0);
    for (gmo_vid i = 0; i < d0; i++) {
      gmo_vid v1 = y0[i];
      if (v1 >= v0) break;
      for (gmo_vid j = 0; j < d0; j++) {
        gmo_vid v2 = y0[j];
        if (v2 >= v1) break;
        counter += gmo_intersect_num_upper(row_of(g, v1), deg_of(g, v1), row_of(g, v2), deg_of(g, v2), v0);
      }
    }
  }
  return counter;
}

/* ---- the other 4-vertex SgL patterns of src/sgl/omp_base.cc:21-31 (round 6).  The loop nests are the reference's; where its innermost
 * loop only counts (`counter += 1` under a filter) the count of the qualifying entries is taken at once -- same number, and a restatement
 * that finishes on R-MAT-14 ---------------------------------------------------------------------------------------------------------- */
uint64_t gmo_3star(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/3star.h:1-13: v1 > v2 > v3, all in N(v0) */
  uint64_t counter = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : counter)
  for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
    gmo_vid d0 = deg_of(g, v0);
    for (gmo_vid i = 0; i < d0; i++)      /* v1 = y0[i] */
      for (gmo_vid j = 0; j < i; j++)     /* v2 = y0[j] < v1 (rows ascending: the `break` at v2 >= v1) */
        counter += (uint64_t)j;           /* the v3 in N(v0) below v2: y0[0 .. j) */
  }
  return counter;
}

uint64_t gmo_4path(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/4path.h:1-14: v0 - v1 - v2 - v3, v2 != v0, v3 < v0 (break), v3 != v1 */
  uint64_t counter = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : counter)
  for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
    const gmo_vid *y0 = row_of(g, v0);
    gmo_vid d0 = deg_of(g, v0);
    for (gmo_vid i = 0; i < d0; i++) {
      gmo_vid v1 = y0[i];
      const gmo_vid *y1 = row_of(g, v1);
      gmo_vid d1 = deg_of(g, v1);
      for (gmo_vid j = 0; j < d1; j++) {
        gmo_vid v2 = y1[j];
        if (v2 == v0) continue;
        const gmo_vid *y2 = row_of(g, v2);
        gmo_vid d2 = deg_of(g, v2);
        /* the v3 in N(v2) with v3 < v0 and v3 != v1: the entries below v0, minus one when v1 is among them (v1 is in N(v2)) */
        gmo_vid below = gmo_bounded(y2, d2, v0); /* (VertexSet::bounded: the entries below v0) */
        counter += (uint64_t)below - (uint64_t)(v1 < v0 ? 1 : 0);
      }
    }
  }
  return counter;
}

uint64_t gmo_tailedtriangle(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/tailedtriangle.h:1-12: v2 in N(v0) ^ N(v1), v2 < v1; v3 in N(v0), v3 != v1, v3 != v2 */
  uint64_t counter = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : counter)
  for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
    const gmo_vid *y0 = row_of(g, v0);
    gmo_vid d0 = deg_of(g, v0);
    for (gmo_vid i = 0; i < d0; i++) {
      gmo_vid v1 = y0[i];
      /* |intersection_set(N(v0), N(v1), v1)| members v2, each with the d0 - 2 entries of N(v0) that are neither v1 nor v2 */
      uint64_t n = gmo_intersect_num_upper(y0, d0, row_of(g, v1), deg_of(g, v1), v1);
      counter += n * (uint64_t)(d0 - 2);
    }
  }
  return counter;
}

uint64_t gmo_house(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/house.h:1-16 */
  uint64_t counter = 0;
#pragma omp parallel reduction(+ : counter)
  {
    gmo_vid *buf = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)(g->max_degree > 0 ? g->max_degree : 1));
#pragma omp for schedule(dynamic, 1)
    for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
      const gmo_vid *y0 = row_of(g, v0);
      gmo_vid d0 = deg_of(g, v0);
      for (gmo_vid i = 0; i < d0; i++) {
        gmo_vid v1 = y0[i];
        if (v1 >= v0) break;
        const gmo_vid *y1 = row_of(g, v1);
        gmo_vid d1 = deg_of(g, v1);
        gmo_vid n = gmo_intersect_set(y0, d0, y1, d1, buf);
        for (gmo_vid p = 0; p < n; p++) {
          gmo_vid v2 = buf[p];
          for (gmo_vid q = 0; q < d1; q++) {
            gmo_vid v3 = y1[q];
            if (v3 == v0 || v3 == v2) continue;
            counter += gmo_intersect_num_except2(y0, d0, row_of(g, v3), deg_of(g, v3), v1, v2);
          }
        }
      }
    }
    free(buf);
  }
  return counter;
}

uint64_t gmo_pentagon(const gmo_graph *g) {
  /* src/sgl/cpu_kernels/pentagon.h:2-17 */
  uint64_t counter = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : counter)
  for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
    const gmo_vid *y0 = row_of(g, v0);
    gmo_vid d0 = deg_of(g, v0);
    for (gmo_vid i = 0; i < d0; i++) {
      gmo_vid v1 = y0[i];
      if (v1 >= v0) break;
      const gmo_vid *y1 = row_of(g, v1);
      gmo_vid d1 = deg_of(g, v1);
      for (gmo_vid j = 0; j < d0; j++) {
        gmo_vid v2 = y0[j];
        if (v2 >= v1) break;
        const gmo_vid *y2 = row_of(g, v2);
        gmo_vid d2 = deg_of(g, v2);
        for (gmo_vid q = 0; q < d2; q++) {
          gmo_vid v3 = y2[q];
          if (v3 >= v0) break;
          if (v3 == v1) continue;
          counter += gmo_intersect_num_upper_except(y1, d1, row_of(g, v3), deg_of(g, v3), v0, v2);
        }
      }
    }
  }
  return counter;
}

/* DFS used for k >= 6 (and cross-checked against the literal k=4,5 loops in the tests):
 * level invariant = candidate set S (ascending) of vertices adjacent (in DAG order) to all chosen ones. */
static uint64_t clique_dfs(const gmo_graph *g, const gmo_vid *S, gmo_vid s, int remaining, gmo_vid *scratch, gmo_vid stride) {
  /* remaining = vertices still to choose INCLUDING the counting level */
  uint64_t c = 0;
  if (remaining == 2) {
    for (gmo_vid i = 0; i < s; i++) c += gmo_intersect_num(S, s, row_of(g, S[i]), deg_of(g, S[i]));
    return c;
  }
  for (gmo_vid i = 0; i < s; i++) {
    gmo_vid n = gmo_intersect_set(S, s, row_of(g, S[i]), deg_of(g, S[i]), scratch);
    if (n) c += clique_dfs(g, scratch, n, remaining - 1, scratch + stride, stride);
  }
  return c;
}

static uint64_t clique_vertex(const gmo_graph *g, int k, gmo_vid v0, gmo_vid *buf, gmo_vid md) {
      /* one iteration of the v0 loop of the automine k-clique solvers */
      const gmo_vid *y0 = row_of(g, v0);
      gmo_vid d0 = deg_of(g, v0);
      uint64_t local = 0;
      if (k == 3) {
        /* automine_3clique, src/clique/cpu_kernels/automine_omp.h:18-30 */
        for (gmo_vid i = 0; i < d0; i++) local += gmo_intersect_num(y0, d0, row_of(g, y0[i]), deg_of(g, y0[i]));
      } else if (k == 4) {
        /* automine_4clique, src/clique/cpu_kernels/automine_omp.h:67-83 */
        for (gmo_vid i = 0; i < d0; i++) {
          gmo_vid v1 = y0[i];
          gmo_vid n = gmo_intersect_set(y0, d0, row_of(g, v1), deg_of(g, v1), buf);
          for (gmo_vid p = 0; p < n; p++) local += gmo_intersect_num(buf, n, row_of(g, buf[p]), deg_of(g, buf[p]));
        }
      } else if (k == 5) {
        /* automine_5clique, src/clique/cpu_kernels/automine_omp.h:138-157 */
        gmo_vid *b2 = buf + md;
        for (gmo_vid i = 0; i < d0; i++) {
          gmo_vid v2 = y0[i];
          gmo_vid n1 = gmo_intersect_set(y0, d0, row_of(g, v2), deg_of(g, v2), buf);
          for (gmo_vid p = 0; p < n1; p++) {
            gmo_vid v3 = buf[p];
            gmo_vid n2 = gmo_intersect_set(buf, n1, row_of(g, v3), deg_of(g, v3), b2);
            for (gmo_vid q = 0; q < n2; q++) local += gmo_intersect_num(b2, n2, row_of(g, b2[q]), deg_of(g, b2[q]));
          }
        }
      } else {
        /* same loop nest, one more intersection level per extra vertex */
        local += clique_dfs(g, y0, d0, k - 1, buf, md);
      }
      return local;
}

static uint64_t clique_strided(const gmo_graph *g, int k, gmo_vid vb, gmo_vid ve, gmo_vid stride, uint64_t *tasks) {
  uint64_t counter = 0, t = 0;
  if (k < 3 || k > 12) return 0;
  if (stride < 1) stride = 1;
  gmo_vid md = g->max_degree > 0 ? g->max_degree : 1;
  gmo_vid n = ve > vb ? (ve - vb + stride - 1) / stride : 0;
#pragma omp parallel reduction(+ : counter, t)
  {
    gmo_vid *buf = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)md * 12);
#pragma omp for schedule(dynamic, 1)
    for (gmo_vid i = 0; i < n; i++) {
      gmo_vid v0 = vb + i * stride;
      t += (uint64_t)deg_of(g, v0);
      counter += clique_vertex(g, k, v0, buf, md);
    }
    free(buf);
  }
  if (tasks) *tasks = t;
  return counter;
}
uint64_t gmo_clique_range(const gmo_graph *g, int k, gmo_vid vb, gmo_vid ve) { return clique_strided(g, k, vb, ve, 1, NULL); }
uint64_t gmo_clique_sample(const gmo_graph *g, int k, gmo_vid stride, gmo_vid offset, uint64_t *tasks) {
  return clique_strided(g, k, offset, g->nv, stride, tasks);
}
uint64_t gmo_clique(const gmo_graph *g, int k) { return gmo_clique_range(g, k, 0, g->nv); }

static void motif3_strided(const gmo_graph *g, gmo_vid vb, gmo_vid ve, gmo_vid stride, uint64_t out[2], uint64_t *tasks) {
  /* automine_3motif, src/motif/cpu_kernels/automine_base.h:2-22; out[0]=wedges, out[1]=triangles */
  uint64_t c0 = 0, c1 = 0, t = 0;
  if (stride < 1) stride = 1;
  gmo_vid n = ve > vb ? (ve - vb + stride - 1) / stride : 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c0, c1, t)
  for (gmo_vid k = 0; k < n; k++) {
    gmo_vid v0 = vb + k * stride;
    const gmo_vid *y0 = row_of(g, v0);
    gmo_vid d0 = deg_of(g, v0);
    gmo_vid f0 = gmo_bounded(y0, d0, v0);
    t += (uint64_t)d0; /* every directed edge is a task (src/motif/gpu_base.cu:34,47) */
    for (gmo_vid i = 0; i < d0; i++) {
      gmo_vid v1 = y0[i];
      c0 += gmo_difference_num_upper(y0, d0, row_of(g, v1), deg_of(g, v1), v1, v1);
    }
    for (gmo_vid i = 0; i < f0; i++) {
      gmo_vid v1 = y0[i];
      c1 += gmo_intersect_num_upper(y0, f0, row_of(g, v1), deg_of(g, v1), v1);
    }
  }
  out[0] = c0;
  out[1] = c1;
  if (tasks) *tasks = t;
}
void gmo_motif3_range(const gmo_graph *g, gmo_vid vb, gmo_vid ve, uint64_t out[2]) { motif3_strided(g, vb, ve, 1, out, NULL); }
void gmo_motif3_sample(const gmo_graph *g, gmo_vid stride, gmo_vid offset, uint64_t out[2], uint64_t *tasks) {
  motif3_strided(g, offset, g->nv, stride, out, tasks);
}
void gmo_motif3(const gmo_graph *g, uint64_t out[2]) { gmo_motif3_range(g, 0, g->nv, out); }

void gmo_motif4(const gmo_graph *g, uint64_t out[6]) {
  /* automine_4motif, src/motif/cpu_kernels/automine_base.h:24-75
   * out: [0] 3-star [1] 4-path [2] tailed-triangle [3] 4-cycle [4] diamond [5] 4-clique */
  uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
  size_t md = (size_t)(g->max_degree > 0 ? g->max_degree : 1);
#pragma omp parallel reduction(+ : c0, c1, c2, c3, c4, c5)
  {
    gmo_vid *mem = (gmo_vid *)malloc(sizeof(gmo_vid) * md * 8);
    gmo_vid *y0n1f1 = mem, *y0y1 = mem + md, *y0f0y1f1 = mem + 2 * md, *n0y1 = mem + 3 * md;
    gmo_vid *y0n1 = mem + 4 * md, *y0f0n1f1 = mem + 5 * md, *n0n1y2 = mem + 6 * md;
#pragma omp for schedule(dynamic, 1)
    for (gmo_vid v0 = 0; v0 < g->nv; v0++) {
      const gmo_vid *y0 = row_of(g, v0);
      gmo_vid d0 = deg_of(g, v0);
      gmo_vid f0 = gmo_bounded(y0, d0, v0);
      for (gmo_vid i1 = 0; i1 < d0; i1++) { /* :31-40 */
        gmo_vid v1 = y0[i1];
        gmo_vid s = gmo_difference_set_upper(y0, d0, row_of(g, v1), deg_of(g, v1), v1, v1, y0n1f1);
        for (gmo_vid i2 = 0; i2 < s; i2++) {
          gmo_vid v2 = y0n1f1[i2];
          c0 += gmo_difference_num_upper(y0n1f1, s, row_of(g, v2), deg_of(g, v2), v2, v2);
        }
      }
      for (gmo_vid i1 = 0; i1 < f0; i1++) { /* :41-72 */
        gmo_vid v1 = y0[i1];
        const gmo_vid *y1 = row_of(g, v1);
        gmo_vid d1 = deg_of(g, v1);
        gmo_vid s_y0y1 = gmo_intersect_set(y0, d0, y1, d1, y0y1);
        gmo_vid s_ff = gmo_intersect_set_upper(y0, f0, y1, d1, v1, y0f0y1f1);
        gmo_vid s_n0y1 = gmo_difference_set(y1, d1, y0, d0, v0, n0y1); /* n0y1 == n0f0y1 (:47-48) */
        gmo_vid s_y0n1 = gmo_difference_set(y0, d0, y1, d1, v1, y0n1);
        gmo_vid s_f0n1f1 = gmo_difference_set_upper(y0, f0, y1, d1, v1, v1, y0f0n1f1);
        for (gmo_vid i2 = 0; i2 < s_y0y1; i2++) { /* :51-56 */
          gmo_vid v2 = y0y1[i2];
          const gmo_vid *y2 = row_of(g, v2);
          gmo_vid d2 = deg_of(g, v2);
          c4 += gmo_difference_num_upper(y0y1, s_y0y1, y2, d2, v2, v2);
          gmo_vid t = gmo_difference_set(y2, d2, y0, d0, v0, n0n1y2);
          c2 += (uint64_t)gmo_difference_set(n0n1y2, t, y1, d1, v1, n0n1y2);
        }
        for (gmo_vid i2 = 0; i2 < s_ff; i2++) { /* :57-61 */
          gmo_vid v2 = y0f0y1f1[i2];
          c5 += gmo_intersect_num_upper(y0f0y1f1, s_ff, row_of(g, v2), deg_of(g, v2), v2);
        }
        for (gmo_vid i2 = 0; i2 < s_y0n1; i2++) { /* :62-66 */
          gmo_vid v2 = y0n1[i2];
          c1 += (uint64_t)gmo_difference_set(n0y1, s_n0y1, row_of(g, v2), deg_of(g, v2), v2, n0n1y2);
        }
        for (gmo_vid i2 = 0; i2 < s_f0n1f1; i2++) { /* :67-71 */
          gmo_vid v2 = y0f0n1f1[i2];
          c3 += gmo_intersect_num_upper(n0y1, s_n0y1, row_of(g, v2), deg_of(g, v2), v0);
        }
      }
    }
    free(mem);
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3; out[4] = c4; out[5] = c5;
}

/* ------------------------------------------------------------------------- */
/* algorithmic bytes                                           */
/* ------------------------------------------------------------------------- */

uint64_t gmo_alg_bytes_tc(const gmo_graph *g) {
  /* B = sum_{(u,v) in E+} 4(d+(u)+d+(v)) + 40|E+| */
  uint64_t b = 0;
#pragma omp parallel for reduction(+ : b) schedule(static)
  for (gmo_vid u = 0; u < g->nv; u++) {
    gmo_vid du = deg_of(g, u);
    const gmo_vid *r = row_of(g, u);
    for (gmo_vid i = 0; i < du; i++) b += 4ull * (uint64_t)(du + deg_of(g, r[i])) + 40ull;
  }
  return b;
}

uint64_t gmo_alg_bytes_diamond(const gmo_graph *g) {
  /* B = sum_{(v0,v1), v1<v0} 4(d(v0)+d(v1)) + 40*ne/2 */
  uint64_t b = 0;
#pragma omp parallel for reduction(+ : b) schedule(static)
  for (gmo_vid u = 0; u < g->nv; u++) {
    gmo_vid du = deg_of(g, u);
    const gmo_vid *r = row_of(g, u);
    for (gmo_vid i = 0; i < du; i++) {
      if (r[i] >= u) break;
      b += 4ull * (uint64_t)(du + deg_of(g, r[i])) + 40ull;
    }
  }
  return b;
}

uint64_t gmo_alg_bytes_clique4(const gmo_graph *g) {
  /* B = sum_e [4(d+(v0)+d+(v1)) + 8|S1| + sum_{v2 in S1}(4(|S1|+d+(v2)) + 16)] + 40|E+| */
  uint64_t b = 0;
#pragma omp parallel reduction(+ : b)
  {
    gmo_vid *buf = (gmo_vid *)malloc(sizeof(gmo_vid) * (size_t)(g->max_degree > 0 ? g->max_degree : 1));
#pragma omp for schedule(dynamic, 64)
    for (gmo_vid u = 0; u < g->nv; u++) {
      gmo_vid du = deg_of(g, u);
      const gmo_vid *r = row_of(g, u);
      for (gmo_vid i = 0; i < du; i++) {
        gmo_vid v = r[i];
        gmo_vid n = gmo_intersect_set(r, du, row_of(g, v), deg_of(g, v), buf);
        b += 4ull * (uint64_t)(du + deg_of(g, v)) + 8ull * (uint64_t)n + 40ull;
        for (gmo_vid p = 0; p < n; p++) b += 4ull * (uint64_t)(n + deg_of(g, buf[p])) + 16ull;
      }
    }
    free(buf);
  }
  return b;
}

uint64_t gmo_alg_bytes_motif3(const gmo_graph *g) {
  /* B = sum_{(v0,v1)} 4(d(v0)+d(v1)) + sum_{v1<v0} 4(d(v0)+d(v1)) + 40*ne */
  uint64_t b = 0;
#pragma omp parallel for reduction(+ : b) schedule(static)
  for (gmo_vid u = 0; u < g->nv; u++) {
    gmo_vid du = deg_of(g, u);
    const gmo_vid *r = row_of(g, u);
    for (gmo_vid i = 0; i < du; i++) {
      uint64_t w = 4ull * (uint64_t)(du + deg_of(g, r[i]));
      b += w + 40ull;
      if (r[i] < u) b += w;
    }
  }
  return b;
}
