// host_graph.h -- C++ host side: the Graph / Pattern surface the reference's mains and solvers use.
//
// Mirrors (names, argument meaning, printed lines, error behaviour) of
//   class Graph    include/graph.h:49-148, src/common/graph.cc:4-124,645-665
//   class Pattern  include/pattern.hh:47-78 (name predicates only; file/label parts are out of scope)
//   Timer          include/timer.h:6-33
// It only loads and holds the CSR; orientation and all mining run on the GPU through the C ABI
// (include/graphminer_amd.h). Nothing here computes on the host.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <sys/time.h>

#include "../../include/graphminer_amd.h"

typedef int32_t vidType;   // include/common.h:36
typedef int64_t eidType;   // include/common.h:37
typedef unsigned long long AccType;

#ifndef USE_DAG
#define USE_DAG 1  // include/defines.h:12-14
#endif

// include/pattern.hh:4-15
static const int num_possible_patterns[] = {0, 1, 1, 2, 6, 21, 112, 853, 11117, 261080};

class Timer {
 public:
  void Start() { gettimeofday(&start_, nullptr); }
  void Stop() { gettimeofday(&stop_, nullptr); }
  double Seconds() const { return (stop_.tv_sec - start_.tv_sec) + 1e-6 * (stop_.tv_usec - start_.tv_usec); }

 private:
  timeval start_{}, stop_{};
};

struct VertexSetView {  // VertexSet(ptr, size, vid) view ctor, include/VertexSet.h:41-42
  const vidType *ptr;
  vidType set_size;
  vidType vid;
  vidType size() const { return set_size; }
  const vidType *begin() const { return ptr; }
  const vidType *end() const { return ptr + set_size; }
};

class Graph {
 public:
  // Graph(prefix, use_dag): loads <prefix>.meta.txt/.vertex.bin/.edge.bin; use_dag -> orientation()
  explicit Graph(const std::string &prefix, bool use_dag = false);
  ~Graph();
  Graph(const Graph &) = delete;             // include/graph.h:59-60
  Graph &operator=(const Graph &) = delete;

  vidType V() const { return n_vertices; }
  eidType E() const { return n_edges; }
  vidType num_vertices() const { return n_vertices; }
  eidType num_edges() const { return n_edges; }
  vidType get_max_degree() const { return max_degree; }
  vidType get_degree(vidType v) const { return vidType(vertices[v + 1] - vertices[v]); }
  const eidType *out_rowptr() const { return vertices.data(); }
  const vidType *out_colidx() const { return edges.data(); }
  VertexSetView N(vidType v) const;  // src/common/graph.cc:172-182
  void print_meta_data() const;      // src/common/graph.cc:645-665
  void orientation();                // src/common/graph.cc:233-279 (runs on GPU 0)
  void sort_neighbors();             // src/common/graph.cc:138-146 (segmented sort on GPU 0)
  gm_csr csr() const { return gm_csr{n_vertices, n_edges, max_degree, vertices.data(), edges.data()}; }
  const std::string &name() const { return name_; }

 private:
  std::string name_, inputfile_path;
  vidType n_vertices = 0, max_degree = 0;
  eidType n_edges = 0;
  int feat_len = 0, num_vertex_classes = 0, num_edge_classes = 0;
  std::vector<eidType> vertices;
  std::vector<vidType> edges;
};

class Pattern {
 public:
  explicit Pattern(std::string name) : name_(std::move(name)) {}
  bool is_diamond() const { return name_ == "diamond"; }
  bool is_rectangle() const { return name_ == "rectangle"; }
  bool is_house() const { return name_ == "house"; }
  bool is_pentagon() const { return name_ == "pentagon"; }
  bool is_tailedtriangle() const { return name_ == "tailedtriangle"; }  // include/pattern.hh:64-66
  bool is_4path() const { return name_ == "4path"; }
  bool is_3star() const { return name_ == "3star"; }
  const std::string &get_name() const { return name_; }

 private:
  std::string name_;
};

// message + exit(1): the reference's reaction to IO / device errors (custom_alloc.h:38-41, cutil_subset.h:4-10)
[[noreturn]] void gm_die(int status, const char *where);

// The link-time solver seam: same signatures as the reference.
void TCSolver(Graph &g, uint64_t &total, int n_gpu, int chunk_size);                       // src/triangle/main.cc:5
void SglSolver(Graph &g, Pattern &p, uint64_t &total, int n_devices, int chunk_size);      // src/sgl/main.cc:7
void CliqueSolver(Graph &g, int k, uint64_t &total, int n_gpu, int chunk_size);            // src/clique/main.cc:6
void MotifSolver(Graph &g, int k, std::vector<uint64_t> &accum, int n_gpu, int chunk_size); // src/motif/main.cc:7
