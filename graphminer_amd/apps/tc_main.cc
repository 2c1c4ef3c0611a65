// tc_gpu_base / tc_multigpu / tc_multigpu_base -- CLI of src/triangle/main.cc:7-27 on the HIP solver.
// argv: <graph> [num_gpu(1)] [chunk_size(1024)] [adj_sorted(1)]; last line "total_num_triangles = N".
#include <cstdlib>
#include <iostream>
#include "../host/graph.h"

int main(int argc, char *argv[]) {
  if (argc < 2) {
    std::cout << "Usage: " << argv[0] << " <graph> [num_gpu(1)] [chunk_size(1024)] [adj_sorted(1)]\n";
    std::cout << "Example: " << argv[0] << " /graph_inputs/mico/graph\n";
    exit(1);
  }
  std::cout << "Triangle Counting: we assume the neighbor lists are sorted.\n";
  Graph g(argv[1], USE_DAG);  // use DAG
  int n_devices = 1;
  int chunk_size = 1024;
  if (argc > 2) n_devices = atoi(argv[2]);
  if (argc > 3) chunk_size = atoi(argv[3]);
  g.print_meta_data();
  if (argc > 4 && !atoi(argv[4])) {
    std::cout << "unsorted neighbor lists are not supported (sort_neighbors, src/common/graph.cc:138, is out of scope)\n";
    exit(1);
  }
#ifndef GM_APP_MULTIGPU
  n_devices = 1;  // *_gpu_base ignores num_gpu (src/triangle/gpu_base.cu:25)
#endif
  uint64_t total = 0;
  TCSolver(g, total, n_devices, chunk_size);
  std::cout << "total_num_triangles = " << total << "\n";
  return 0;
}
