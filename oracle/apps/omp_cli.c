/*
 * omp_cli.c -- the *_omp_base command-line surface on top of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY (see ../gm_oracle.h). Compiled four
 * times with -DGMO_APP_TC / _SGL / _CLIQUE / _MOTIF into oracle/bin/
 * {tc,sgl,clique,kcl,motif}_omp_base. BASELINE.json config 1 ("tc_omp_base on
 * inputs/citeseer/graph, CPU-only plumbing") is this binary.
 *
 * argv order, defaults and the FINAL result line are byte-identical to the
 * reference mains (src/triangle/main.cc:7-27, src/sgl/main.cc:9-35,
 * src/clique/main.cc:8-28, src/motif/main.cc:9-31; Pangolin spelling
 * src/pangolin/clique/main.cc:8,20 when built with -DGMO_KCL_SPELLING).
 */
#include "../gm_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

static double now_sec(void) { /* Timer, include/timer.h:6-33 (gettimeofday wall clock) */
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

static void print_path_line(const char *prefix) {
  /* src/common/graph.cc:13-17 */
  char path[4096], name[4096];
  path[0] = name[0] = 0;
  const char *s = strrchr(prefix, '/');
  if (s) { size_t n = (size_t)(s - prefix); memcpy(path, prefix, n); path[n] = 0; }
  const char *t = strrchr(path, '/');
  if (t) strcpy(name, t + 1);
  printf("input file path: %s, graph name: %s\n", path, name);
}

static void print_meta(const gmo_graph *g) {
  /* Graph::print_meta_data, src/common/graph.cc:645-665 (label/feature lines are free-form extras) */
  printf("|V|: %d, |E|: %lld, Max Degree: %d\n", g->nv, (long long)g->ne, g->max_degree);
}

static int load_or_die(const char *prefix, gmo_graph *g) {
  print_path_line(prefix);
  int rc = gmo_load(prefix, g);
  if (rc) { fprintf(stderr, "cannot load graph '%s' (error %d)\n", prefix, rc); exit(1); }
  return 0;
}

__attribute__((unused)) static void orient_in_place(gmo_graph *g) {
  printf("Orientation enabled, using DAG\n"); /* graph.cc:234 */
  double t0 = now_sec();
  gmo_graph dag;
  gmo_orient(g, &dag);
  gmo_free(g);
  *g = dag;
  printf("Time on generating the DAG: %g sec\n", now_sec() - t0); /* graph.cc:278 */
}

int main(int argc, char **argv) {
  gmo_graph g;
#if defined(GMO_APP_TC)
  if (argc < 2) {
    printf("Usage: %s <graph> [num_gpu(1)] [chunk_size(1024)] [adj_sorted(1)]\n", argv[0]);
    printf("Example: %s /graph_inputs/mico/graph\n", argv[0]);
    exit(1);
  }
  printf("Triangle Counting: we assume the neighbor lists are sorted.\n");
  load_or_die(argv[1], &g);
  orient_in_place(&g);
  print_meta(&g);
  printf("OpenMP TC (%d threads)\n", gmo_num_threads());
  double t0 = now_sec();
  uint64_t total = gmo_tc(&g);
  printf("runtime [omp_base] = %g sec\n", now_sec() - t0);
  printf("total_num_triangles = %llu\n", (unsigned long long)total);
#elif defined(GMO_APP_SGL)
  if (argc < 3) {
    fprintf(stderr, "usage: %s <graph prefix> <pattern> [num_gpu(1)] [chunk_size(1024)]\n", argv[0]);
    printf("Example: %s /graph_inputs/mico/graph rectangle\n", argv[0]);
    exit(1);
  }
  printf("Subgraph Listing/Counting (undirected graph only)\n");
  load_or_die(argv[1], &g);
  printf("Pattern: %s\n", argv[2]);
  print_meta(&g);
  printf("OpenMP edge-induced subgraph listing (%d threads) ...\n", gmo_num_threads());
  double t0 = now_sec();
  uint64_t total = 0;
  if (!strcmp(argv[2], "diamond")) total = gmo_diamond(&g);
  else if (!strcmp(argv[2], "rectangle")) total = gmo_rectangle(&g);
  else if (!strcmp(argv[2], "house")) total = gmo_house(&g);
  else if (!strcmp(argv[2], "pentagon")) total = gmo_pentagon(&g);
  else printf("Not implemented\n"); /* src/sgl/omp_base.cc:51-53 */
  printf("runtime = %g seconds\n", now_sec() - t0);
  printf("total_num = %llu\n", (unsigned long long)total);
#elif defined(GMO_APP_CLIQUE)
  if (argc < 3) {
    printf("Usage: %s<graph> <k> [ngpu(0)] [chunk_size(1024)]\n", argv[0]);
    printf("Example: %s /graph_inputs/mico/graph 4\n", argv[0]);
    exit(1);
  }
  printf("k-clique listing with undirected graphs\n");
  printf("Using DAG (static orientation)\n");
  load_or_die(argv[1], &g);
  orient_in_place(&g);
  int k = atoi(argv[2]);
  print_meta(&g);
  printf("OpenMP %d-clique listing (%d threads)\n", k, gmo_num_threads());
  if (k < 3 || k > 8) { printf("Not implemented yet\n"); exit(0); } /* automine_omp.h:179-182 (k>5 there) */
  double t0 = now_sec();
  uint64_t total = gmo_clique(&g, k);
  printf("runtime [omp_base] = %g sec\n", now_sec() - t0);
#ifdef GMO_KCL_SPELLING
  printf("\ntotal_num_cliques = %llu\n\n", (unsigned long long)total); /* src/pangolin/clique/main.cc:20 */
#else
  printf("num_%d-cliques = %llu\n", k, (unsigned long long)total);
#endif
#elif defined(GMO_APP_MOTIF)
  if (argc < 3) {
    printf("Usage: %s<graph> <k> [ngpu(0)] [chunk_size(1024)]\n", argv[0]);
    printf("Example: %s /graph_inputs/mico/graph 4\n", argv[0]);
    exit(1);
  }
  load_or_die(argv[1], &g);
  int k = atoi(argv[2]);
  printf("%d-motif counting (only for undirected graphs)\n", k);
  print_meta(&g);
  int np = k == 3 ? 2 : (k == 4 ? 6 : 0); /* num_possible_patterns, include/pattern.hh:4-15 */
  printf("num_patterns: %d\n", np);
  printf("OpenMP Motif solver (%d threads) ...\n", gmo_num_threads());
  if (np == 0) { printf("Not implemented yet\n"); exit(0); }
  uint64_t out[6] = {0, 0, 0, 0, 0, 0};
  double t0 = now_sec();
  if (k == 3) gmo_motif3(&g, out); else gmo_motif4(&g, out);
  printf("runtime [omp_base] = %g\n", now_sec() - t0);
  for (int i = 0; i < np; i++) printf("pattern %d: %llu\n", i, (unsigned long long)out[i]);
#else
#error "define one of GMO_APP_TC / GMO_APP_SGL / GMO_APP_CLIQUE / GMO_APP_MOTIF"
#endif
  gmo_free(&g);
  return 0;
}
