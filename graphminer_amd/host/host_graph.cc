// host_graph.cc -- loader + meta printing + GPU orientation for the C++ host side (see host_graph.h).
#include "host_graph.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

[[noreturn]] void gm_die(int status, const char *where) {
  // the reference has no error returns: CUDA_SAFE_CALL -> fprintf + exit(EXIT_FAILURE) (cutil_subset.h:4-10)
  std::fprintf(stderr, "error: %s: %s", where, gm_strerror(status));
  const char *d = gm_last_error();
  if (d && *d) std::fprintf(stderr, " [%s]", d);
  std::fprintf(stderr, "\n");
  std::exit(EXIT_FAILURE);
}

template <typename T>
static void read_file(const std::string &fname, std::vector<T> &out, size_t n) {
  // read_file, include/custom_alloc.h:34-44: open failure -> message + exit(1)
  out.resize(n);
  std::ifstream f(fname, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "Error opening file %s\n", fname.c_str());
    std::exit(1);
  }
  f.read(reinterpret_cast<char *>(out.data()), std::streamsize(sizeof(T) * n));
  if (size_t(f.gcount()) != sizeof(T) * n) {
    std::fprintf(stderr, "Error reading file %s (short read)\n", fname.c_str());
    std::exit(1);
  }
}

Graph::Graph(const std::string &prefix, bool use_dag) {
  // file-name parsing and the first printed line: src/common/graph.cc:13-17
  size_t i = prefix.rfind('/');
  if (i != std::string::npos) inputfile_path = prefix.substr(0, i);
  i = inputfile_path.rfind('/');
  if (i != std::string::npos) name_ = inputfile_path.substr(i + 1);
  std::cout << "input file path: " << inputfile_path << ", graph name: " << name_ << "\n";

  std::ifstream f_meta(prefix + ".meta.txt");  // graph.cc:21-35
  if (!f_meta) {
    std::fprintf(stderr, "Error opening file %s.meta.txt\n", prefix.c_str());
    std::exit(1);
  }
  long long nv = 0, ne = 0;
  int vid_size = 0, eid_size = 0, vlabel_size = 0, elabel_size = 0, md = 0;
  f_meta >> nv >> ne >> vid_size >> eid_size >> vlabel_size >> elabel_size >> md >> feat_len >> num_vertex_classes >>
      num_edge_classes;
  if (vid_size != int(sizeof(vidType)) || eid_size != int(sizeof(eidType)) || !(md > 0 && md < nv)) {
    // the reference asserts here (graph.cc:30-34); assert -> abort
    std::fprintf(stderr, "bad graph meta data: need sizeof(vid)==4, sizeof(eid)==8, 0 < max_degree < nv\n");
    std::abort();
  }
  n_vertices = vidType(nv);
  n_edges = ne;
  max_degree = md;
  read_file(prefix + ".vertex.bin", vertices, size_t(nv) + 1);  // graph.cc:38
  read_file(prefix + ".edge.bin", edges, size_t(ne));           // graph.cc:41
  if (use_dag) orientation();                                    // graph.cc:117-120
}

Graph::~Graph() = default;

VertexSetView Graph::N(vidType v) const {
  const eidType begin = vertices[v], end = vertices[v + 1];
  if (begin > end) {  // graph.cc:176-179
    std::fprintf(stderr, "vertex %u bounds error: [%lu, %lu)\n", unsigned(v), (unsigned long)begin, (unsigned long)end);
    std::exit(1);
  }
  return VertexSetView{edges.data() + begin, vidType(end - begin), v};
}

void Graph::sort_neighbors() {
  std::cout << "Sorting the neighbor lists (used for pattern mining)\n";  // graph.cc:139
  gm_csr h = csr();
  gm_graph *dg = nullptr;
  int rc = gm_graph_upload(&h, 0, &dg);
  if (rc) gm_die(rc, "gm_graph_upload");
  rc = gm_graph_sort_neighbors(dg);
  if (rc) gm_die(rc, "gm_graph_sort_neighbors");
  rc = gm_graph_download(dg, nullptr, edges.data());
  if (rc) gm_die(rc, "gm_graph_download");
  gm_graph_free(dg);
}

void Graph::orientation() {
  std::cout << "Orientation enabled, using DAG\n";  // graph.cc:234
  Timer t;
  t.Start();
  gm_csr h = csr();
  gm_graph *sym = nullptr, *dag = nullptr;
  int rc = gm_graph_upload(&h, 0, &sym);
  if (rc) gm_die(rc, "gm_graph_upload");
  rc = gm_graph_orient(sym, &dag);
  if (rc) gm_die(rc, "gm_graph_orient");
  gm_csr m;
  gm_graph_meta(dag, &m);
  std::vector<eidType> nrp(size_t(m.nv) + 1);
  std::vector<vidType> nci(size_t(m.ne));
  rc = gm_graph_download(dag, nrp.data(), nci.data());
  if (rc) gm_die(rc, "gm_graph_download");
  gm_graph_free(sym);
  gm_graph_free(dag);
  vertices.swap(nrp);
  edges.swap(nci);
  n_edges = m.ne;
  max_degree = m.max_deg;  // graph.cc:252
  t.Stop();
  std::cout << "Time on generating the DAG: " << t.Seconds() << " sec\n";  // graph.cc:278
}

void Graph::print_meta_data() const {
  // src/common/graph.cc:645-665
  std::cout << "|V|: " << n_vertices << ", |E|: " << n_edges << ", Max Degree: " << max_degree << "\n";
  if (num_vertex_classes > 0) std::cout << "vertex-|Σ|: " << num_vertex_classes << "\n";
  else std::cout << "This graph does not have vertex labels\n";
  if (num_edge_classes > 0) std::cout << "edge-|Σ|: " << num_edge_classes << "\n";
  else std::cout << "This graph does not have edge labels\n";
  if (feat_len > 0) std::cout << "Vertex feature vector length: " << feat_len << "\n";
  else std::cout << "This graph has no input vertex features\n";
}
