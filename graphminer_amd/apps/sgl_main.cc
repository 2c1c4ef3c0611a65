// sgl_gpu_base / sgl_multigpu -- CLI of src/sgl/main.cc:9-35.
// argv: <graph prefix> <pattern> [num_gpu(1)] [chunk_size(1024)]; prints "Pattern: <name>", "total_num = N".
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include "../host/graph.h"

int main(int argc, char **argv) {
  if (argc < 3) {
    std::cerr << "usage: " << argv[0] << " <graph prefix> <pattern> [num_gpu(1)] [chunk_size(1024)]\n";
    printf("Example: %s /graph_inputs/mico/graph rectangle\n", argv[0]);
    exit(1);
  }
  std::cout << "Subgraph Listing/Counting (undirected graph only)\n";
  Graph g(argv[1]);
  Pattern patt(argv[2]);
  std::cout << "Pattern: " << patt.get_name() << "\n";
  int n_devices = 1;
  int chunk_size = 1024;
  if (argc > 3) n_devices = atoi(argv[3]);
  if (argc > 4) chunk_size = atoi(argv[4]);
  g.print_meta_data();
#ifndef GM_APP_MULTIGPU
  n_devices = 1;
#endif
  uint64_t h_total = 0;
  SglSolver(g, patt, h_total, n_devices, chunk_size);
  std::cout << "total_num = " << h_total << "\n";
  return 0;
}
