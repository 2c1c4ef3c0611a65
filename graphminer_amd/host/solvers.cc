// solvers.cc -- the four link-time solver entry points of the reference, as thin shims over the C ABI.
//
//   TCSolver      src/triangle/gpu_base.cu:25-72, multi-GPU src/triangle/multigpu_base.cu:25-105
//   SglSolver     src/sgl/gpu_base.cu:21-103
//   CliqueSolver  src/clique/gpu_base.cu:14-80,  multi-GPU src/clique/multigpu.cu:20-139
//   MotifSolver   src/motif/gpu_base.cu:21-110
//
// The work (one GPU: upload + gm_* call; n GPUs: broadcast, shares, one all-reduce) is host/multi.cc, which takes the CSR as a plain
// gm_csr: the same runner links under the reference's own Graph class (integration/hip_solvers.cc).
#include "host_graph.h"
#include "multi.h"

#include <cstdlib>
#include <iostream>

using gmhost::Job;
static bool run(Graph &g, const Job &j, int n_gpu, int chunk, uint64_t *out) { return gmhost::run(g.csr(), j, n_gpu, chunk, out); }


void TCSolver(Graph &g, uint64_t &total, int n_gpu, int chunk_size) {
  Job j;
  j.kind = Job::TC;
  uint64_t out[8] = {0};
  run(g, j, n_gpu, chunk_size, out);
  total += out[0];  // multi-GPU solvers '+=' into the caller-zeroed total (src/triangle/multigpu.cu:84)
}

void SglSolver(Graph &g, Pattern &p, uint64_t &total, int n_devices, int chunk_size) {
  Job j;
  j.kind = Job::SGL;
  j.pattern = p.get_name().c_str();
  uint64_t out[8] = {0};
  if (!run(g, j, n_devices, chunk_size, out)) {
    std::cout << "Not implemented\n";  // src/sgl/omp_base.cc:51-53: total stays 0
    return;
  }
  total += out[0];
}

void CliqueSolver(Graph &g, int k, uint64_t &total, int n_gpu, int chunk_size) {
  if (k < 3 || k > GM_MAX_CLIQUE_K) {  // (the reference's solvers stop at 8: automine_omp.h:179-182, gpu_base.cu:59-71; its generic ones do not)
    std::cout << "Not implemented yet\n";  // src/clique/cpu_kernels/automine_omp.h:179-182
    std::exit(0);
  }
  Job j;
  j.kind = Job::CLIQUE;
  j.k = k;
  uint64_t out[8] = {0};
  if (!run(g, j, n_gpu, chunk_size, out)) {
    std::cout << "Not supported right now\n";  // src/clique/gpu_base.cu:70
    return;
  }
  total += out[0];
}

void MotifSolver(Graph &g, int k, std::vector<uint64_t> &accum, int n_gpu, int chunk_size) {
  Job j;
  j.kind = Job::MOTIF;
  j.k = k;
  j.ncounts = int(accum.size());
  if (k != 3 && k != 4) {
    std::cout << "Not supported right now\n";  // src/motif/gpu_base.cu:101
    return;
  }
  uint64_t out[8] = {0};
  if (!run(g, j, n_gpu, chunk_size, out)) {
    std::cout << "Not supported right now\n";
    return;
  }
  for (size_t i = 0; i < accum.size() && i < 8; ++i) accum[i] += out[i];
}
