// hip_solvers.cc -- the solver objects a GraphMiner maintainer links INSTEAD of omp_base.o / gpu_base.o / multigpu.o
// (INTEGRATION.md section 2). Compiled against the REFERENCE's own headers (include/graph.h, include/pattern.hh) and
// linked with the reference's own main.cc + graph.cc + VertexSet.cc by oracle/ref/Makefile into
// oracle/_ref/{tc,sgl,clique,motif}_hip_base and {tc,clique}_hip_multigpu -- the drop-in is exercised end to end, reference
// loader and CLI included. Exactly one of GM_SHIM_TC / GM_SHIM_SGL / GM_SHIM_CLIQUE / GM_SHIM_MOTIF is defined per object.
//
// n_gpu (argv[2] / argv[3] of the reference's mains, src/triangle/main.cc:14, src/clique/main.cc:17) is HONOURED: the runner
// (graphminer_amd/host/multi.cc, which takes a plain gm_csr) uploads to one GPU, or replicates the CSR with ncclBroadcast, gives every
// device its share of the task chunks and combines the counts with ONE ncclAllReduce(ncclUint64) -- the seam of the reference's
// `clique_multigpu` (src/clique/multigpu.cu:20: void CliqueSolver(Graph &g, int k, uint64_t &total, int n_gpus, int chunk_size)).
#include "graph.h"
#if defined(GM_SHIM_SGL) || defined(GM_SHIM_MOTIF)
#include "pattern.hh"
#endif
#include "graphminer_amd.h"
#include "../graphminer_amd/host/multi.h"

using gmhost::Job;

static bool run(Graph &g, const Job &j, int n_gpu, int chunk, uint64_t *out) {
  // exactly what Graph::out_rowptr() / out_colidx() / V() / E() / get_max_degree() expose (include/graph.h:63-64,73,83-84)
  const gm_csr h = {g.V(), (int64_t)g.E(), g.get_max_degree(), (const int64_t *)g.out_rowptr(), (const int32_t *)g.out_colidx()};
  return gmhost::run(h, j, n_gpu, chunk, out);
}

#if defined(GM_SHIM_TC)
void TCSolver(Graph &g, uint64_t &total, int n_gpu, int chunk_size) {  // g is already oriented by the reference's Graph ctor
  Job j;
  j.kind = Job::TC;
  j.name = "hip_base";
  uint64_t out[8] = {0};
  run(g, j, n_gpu, chunk_size, out);
  total += out[0];  // the multi-GPU solvers '+=' into the caller-zeroed total (src/triangle/multigpu.cu:84)
}
#elif defined(GM_SHIM_SGL)
void SglSolver(Graph &g, Pattern &p, uint64_t &total, int n_devices, int chunk_size) {
  Job j;
  j.kind = Job::SGL;
  j.name = "hip_base";
  const std::string name = p.get_name();
  j.pattern = name.c_str();
  uint64_t out[8] = {0};
  if (!run(g, j, n_devices, chunk_size, out)) {
    std::cout << "Not implemented\n";  // src/sgl/omp_base.cc:51-53: total stays 0
    return;
  }
  total += out[0];
}
#elif defined(GM_SHIM_CLIQUE)
void CliqueSolver(Graph &g, int k, uint64_t &total, int n_gpu, int chunk_size) {
  Job j;
  j.kind = Job::CLIQUE;
  j.k = k;
  j.name = "hip_base";
  uint64_t out[8] = {0};
  if (!run(g, j, n_gpu, chunk_size, out)) {
    std::cout << "Not supported right now\n";  // src/clique/gpu_base.cu:70
    return;
  }
  total += out[0];
}
#elif defined(GM_SHIM_MOTIF)
void MotifSolver(Graph &g, int k, std::vector<uint64_t> &accum, int n_gpu, int chunk_size) {
  Job j;
  j.kind = Job::MOTIF;
  j.k = k;
  j.ncounts = (int)accum.size();
  j.name = "hip_base";
  uint64_t out[8] = {0};
  if (!run(g, j, n_gpu, chunk_size, out)) {
    std::cout << "Not supported right now\n";  // src/motif/gpu_base.cu:101
    return;
  }
  for (size_t i = 0; i < accum.size() && i < 8; ++i) accum[i] += out[i];
}
#endif
