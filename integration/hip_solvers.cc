// hip_solvers.cc -- the solver objects a GraphMiner maintainer links INSTEAD of omp_base.o / gpu_base.o
// (INTEGRATION.md section 2). Compiled against the REFERENCE's own headers (include/graph.h, include/pattern.hh) and
// linked with the reference's own main.cc + graph.cc + VertexSet.cc by oracle/ref/Makefile into
// oracle/_ref/{tc,sgl,clique,motif}_hip_base -- the drop-in is exercised end to end, reference loader and CLI included.
// Exactly one of GM_SHIM_TC / GM_SHIM_SGL / GM_SHIM_CLIQUE / GM_SHIM_MOTIF is defined per object.
#include "graph.h"
#if defined(GM_SHIM_SGL) || defined(GM_SHIM_MOTIF)
#include "pattern.hh"
#endif
#include "graphminer_amd.h"

static void gm_or_die(int rc, const char *what) {
  if (rc == GM_OK) return;
  fprintf(stderr, "%s: %s [%s]\n", what, gm_strerror(rc), gm_last_error());
  exit(EXIT_FAILURE);  // same reaction as CUDA_SAFE_CALL (include/cutil_subset.h:4-10)
}

static gm_graph *upload(Graph &g) {
  gm_csr h = {g.V(), (int64_t)g.E(), g.get_max_degree(), (const int64_t *)g.out_rowptr(), (const int32_t *)g.out_colidx()};
  gm_graph *dg = nullptr;
  gm_or_die(gm_graph_upload(&h, 0, &dg), "gm_graph_upload");
  return dg;
}

static void report(const char *name, const gm_stats &st) {
  std::cout << "runtime [" << name << "] = " << st.kernel_ms * 1e-3 << " sec\n";
  std::cout << "throughput = " << double(st.tasks) / (st.kernel_ms * 1e-3) / 1e9 << " billion Traversed Edges Per Second (TEPS)\n";
}

#if defined(GM_SHIM_TC)
void TCSolver(Graph &g, uint64_t &total, int, int chunk_size) {  // g is already oriented by the reference's Graph ctor
  gm_graph *dg = upload(g);
  gm_launch la = {};
  la.chunk = chunk_size > 0 ? chunk_size : 0;  // the reference main passes its argv value, default 1024: honoured
  gm_stats st = {};
  uint64_t count = 0;
  gm_or_die(gm_tc(dg, &la, &count, &st), "gm_tc");
  report("hip_base", st);
  total = count;
  gm_graph_free(dg);
}
#elif defined(GM_SHIM_SGL)
void SglSolver(Graph &g, Pattern &p, uint64_t &total, int, int chunk_size) {
  gm_graph *dg = upload(g);
  gm_launch la = {};
  la.chunk = chunk_size > 0 ? chunk_size : 0;  // the reference main passes its argv value, default 1024: honoured
  gm_stats st = {};
  uint64_t count = 0;
  int rc = gm_sgl(dg, p.get_name().c_str(), &la, &count, &st);
  if (rc == GM_ERR_UNSUPPORTED) std::cout << "Not implemented\n";  // src/sgl/omp_base.cc:51-53
  else { gm_or_die(rc, "gm_sgl"); report("hip_base", st); }
  total = count;
  gm_graph_free(dg);
}
#elif defined(GM_SHIM_CLIQUE)
void CliqueSolver(Graph &g, int k, uint64_t &total, int, int chunk_size) {
  gm_graph *dg = upload(g);
  gm_launch la = {};
  la.chunk = chunk_size > 0 ? chunk_size : 0;  // the reference main passes its argv value, default 1024: honoured
  gm_stats st = {};
  uint64_t count = 0;
  gm_or_die(gm_clique(dg, k, &la, &count, &st), "gm_clique");
  report("hip_base", st);
  total = count;
  gm_graph_free(dg);
}
#elif defined(GM_SHIM_MOTIF)
void MotifSolver(Graph &g, int k, std::vector<uint64_t> &accum, int, int chunk_size) {
  gm_graph *dg = upload(g);
  gm_launch la = {};
  la.chunk = chunk_size > 0 ? chunk_size : 0;  // the reference main passes its argv value, default 1024: honoured
  gm_stats st = {};
  std::vector<uint64_t> c(accum.size(), 0);
  gm_or_die(gm_motif(dg, k, &la, c.data(), (int)c.size(), &st), "gm_motif");
  report("hip_base", st);
  for (size_t i = 0; i < accum.size(); ++i) accum[i] += c[i];
  gm_graph_free(dg);
}
#endif
