// motif_gpu_base / motif_multigpu -- CLI of src/motif/main.cc:9-31.
// argv: <graph> <k> [ngpu] [chunk_size(1024)]; prints "pattern i: N" for i < num_possible_patterns[k].
#include <cstdlib>
#include <iostream>
#include <vector>
#include "../host/graph.h"

int main(int argc, char *argv[]) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << "<graph> <k> [ngpu(0)] [chunk_size(1024)]\n";
    std::cout << "Example: " << argv[0] << " /graph_inputs/mico/graph 4\n";
    exit(1);
  }
  Graph g(argv[1]);
  int k = atoi(argv[2]);
  int n_devices = 1;
  int chunk_size = 1024;
  if (argc > 3) n_devices = atoi(argv[3]);
  if (argc > 4) chunk_size = atoi(argv[4]);
  std::cout << k << "-motif counting (only for undirected graphs)\n";
  g.print_meta_data();
  if (k < 0 || k > 9) exit(1);
  int num_patterns = num_possible_patterns[k];
  std::cout << "num_patterns: " << num_patterns << "\n";
  std::vector<uint64_t> total(num_patterns, 0);
#ifndef GM_APP_MULTIGPU
  n_devices = 1;
#endif
  MotifSolver(g, k, total, n_devices, chunk_size);
  for (int i = 0; i < num_patterns; i++) std::cout << "pattern " << i << ": " << total[i] << "\n";
  return 0;
}
