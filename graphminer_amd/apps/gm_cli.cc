// gm_cli.cc -- one source for the whole command-line surface of the HIP solvers.
//
// Built with -DGM_APP=<TC|SGL|CLIQUE|MOTIF> (+ -DGM_APP_MULTIGPU, + -DGM_KCL_SPELLING) into
//   tc_gpu_base tc_multigpu tc_multigpu_base | sgl_gpu_base sgl_multigpu | clique_gpu_base clique_multigpu kcl_gpu_base |
//   motif_gpu_base motif_multigpu
// The observable behaviour -- positional argv, defaults, usage text, banner and FINAL result lines -- is that of the
// reference mains (src/triangle/main.cc:7-27, src/sgl/main.cc:9-35, src/clique/main.cc:8-28, src/motif/main.cc:9-31,
// Pangolin spelling src/pangolin/clique/main.cc:20); scripts that grep those lines keep working.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/graphminer_amd.h"
#include "../host/host_graph.h"

#define GM_TC 1
#define GM_SGL 2
#define GM_CLIQUE 3
#define GM_MOTIF 4
#ifndef GM_APP
#error "compile with -DGM_APP=GM_TC|GM_SGL|GM_CLIQUE|GM_MOTIF"
#endif

namespace {

struct Cli {
  std::string graph;   // <graph> prefix
  std::string second;  // <pattern> (sgl) or <k> (clique, motif)
  int n_gpu = 1;       // [num_gpu(1)]
  int chunk = 0;       // [chunk_size(1024)]: honoured when given; omitted = 0 = the library default (<= 1024 task edges, adapted to the rank count)
  int adj_sorted = 1;  // [adj_sorted(1)] (tc only)
};

// positional layout: TC has no <second>; the others do
Cli parse(int argc, char **argv, bool has_second) {
  Cli c;
  int i = 1;
  c.graph = argv[i++];
  if (has_second) c.second = argv[i++];
  if (i < argc) c.n_gpu = std::atoi(argv[i++]);
  if (i < argc) c.chunk = std::atoi(argv[i++]);
  if (i < argc) c.adj_sorted = std::atoi(argv[i++]);
#ifndef GM_APP_MULTIGPU
  c.n_gpu = 1;  // *_gpu_base binaries ignore num_gpu, as the reference's gpu_base solvers do
#endif
  return c;
}

void usage_and_exit(const char *self) {
#if GM_APP == GM_TC
  std::printf("Usage: %s <graph> [num_gpu(1)] [chunk_size(1024)] [adj_sorted(1)]\n", self);
  std::printf("Example: %s /graph_inputs/mico/graph\n", self);
#elif GM_APP == GM_SGL
  std::fprintf(stderr, "usage: %s <graph prefix> <pattern> [num_gpu(1)] [chunk_size(1024)]\n", self);
  std::printf("Example: %s /graph_inputs/mico/graph rectangle\n", self);
#else
  std::printf("Usage: %s<graph> <k> [ngpu(0)] [chunk_size(1024)]\n", self);
  std::printf("Example: %s /graph_inputs/mico/graph 4\n", self);
#endif
  std::exit(1);
}

}  // namespace

// `--dev NAME=VALUE` anywhere on the command line sets a developer option of the library (gm_dev_option, include/graphminer_amd.h: the
// switches tests use to reach a path on a small graph) and is removed from argv before the reference's positional layout is read.
// Nothing is read from the environment.
static int strip_dev_options(int argc, char **argv) {
  int n = 0;
  for (int i = 0; i < argc; ++i) {
    if (std::string(argv[i]) == "--dev" && i + 1 < argc) {
      const std::string kv = argv[++i];
      const size_t eq = kv.find('=');
      const int rc = eq == std::string::npos ? gm_dev_option(kv.c_str(), "1") : gm_dev_option(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str());
      if (rc != GM_OK) {
        std::fprintf(stderr, "bad --dev option: %s\n", kv.c_str());
        std::exit(1);
      }
      continue;
    }
    argv[n++] = argv[i];
  }
  return n;
}

int main(int argc, char **argv) {
  argc = strip_dev_options(argc, argv);
  const bool has_second = (GM_APP != GM_TC);
  if (argc < (has_second ? 3 : 2)) usage_and_exit(argv[0]);
  const Cli c = parse(argc, argv, has_second);

#if GM_APP == GM_TC
  std::printf("Triangle Counting: we assume the neighbor lists are sorted.\n");
  std::fflush(stdout);
  Graph g(c.graph, USE_DAG);
  g.print_meta_data();
  if (!c.adj_sorted) g.sort_neighbors();  // src/triangle/main.cc:22
  uint64_t total = 0;
  TCSolver(g, total, c.n_gpu, c.chunk);
  std::printf("total_num_triangles = %llu\n", (unsigned long long)total);

#elif GM_APP == GM_SGL
  std::printf("Subgraph Listing/Counting (undirected graph only)\n");
  std::fflush(stdout);
  Graph g(c.graph);
  Pattern patt(c.second);
  std::printf("Pattern: %s\n", patt.get_name().c_str());
  std::fflush(stdout);
  g.print_meta_data();
  uint64_t total = 0;
  SglSolver(g, patt, total, c.n_gpu, c.chunk);
  std::printf("total_num = %llu\n", (unsigned long long)total);

#elif GM_APP == GM_CLIQUE
  std::printf("k-clique listing with undirected graphs\n");
  if (USE_DAG) std::printf("Using DAG (static orientation)\n");
  std::fflush(stdout);
  Graph g(c.graph, USE_DAG);
  const int k = std::atoi(c.second.c_str());
  g.print_meta_data();
  uint64_t total = 0;
  CliqueSolver(g, k, total, c.n_gpu, c.chunk);
#ifdef GM_KCL_SPELLING
  std::printf("\ntotal_num_cliques = %llu\n\n", (unsigned long long)total);
#else
  std::printf("num_%d-cliques = %llu\n", k, (unsigned long long)total);
#endif

#elif GM_APP == GM_MOTIF
  Graph g(c.graph);
  const int k = std::atoi(c.second.c_str());
  std::printf("%d-motif counting (only for undirected graphs)\n", k);
  std::fflush(stdout);
  g.print_meta_data();
  if (k < 0 || k > 9) return 1;
  const int np = num_possible_patterns[k];
  std::printf("num_patterns: %d\n", np);
  std::fflush(stdout);
  std::vector<uint64_t> counts((size_t)np, 0);
  MotifSolver(g, k, counts, c.n_gpu, c.chunk);
  for (int i = 0; i < np; ++i) std::printf("pattern %d: %llu\n", i, (unsigned long long)counts[(size_t)i]);
#endif
  return 0;
}
