// clique_gpu_base / clique_multigpu (and the Pangolin spelling kcl_gpu_base) -- CLI of src/clique/main.cc:8-28.
// argv: <graph> <k> [ngpu] [chunk_size(1024)]; last line "num_<k>-cliques = N"
// (-DGM_KCL_SPELLING: "total_num_cliques = N", src/pangolin/clique/main.cc:20).
#include <cstdlib>
#include <iostream>
#include "../host/graph.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << "<graph> <k> [ngpu(0)] [chunk_size(1024)]\n";
    std::cout << "Example: " << argv[0] << " /graph_inputs/mico/graph 4\n";
    exit(1);
  }
  std::cout << "k-clique listing with undirected graphs\n";
  if (USE_DAG) std::cout << "Using DAG (static orientation)\n";
  Graph g(argv[1], USE_DAG);  // use DAG
  int k = atoi(argv[2]);
  int n_devices = 1;
  int chunk_size = 1024;
  if (argc > 3) n_devices = atoi(argv[3]);
  if (argc > 4) chunk_size = atoi(argv[4]);
  g.print_meta_data();
#ifndef GM_APP_MULTIGPU
  n_devices = 1;
#endif
  uint64_t total = 0;
  CliqueSolver(g, k, total, n_devices, chunk_size);
#ifdef GM_KCL_SPELLING
  std::cout << "\ntotal_num_cliques = " << total << "\n\n";
#else
  std::cout << "num_" << k << "-cliques = " << total << "\n";
#endif
  return 0;
}
