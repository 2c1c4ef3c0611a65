// multi.cc -- the one-GPU / n-GPU runner behind the four link-time solver entry points (host/multi.h).
//
//   TCSolver      src/triangle/gpu_base.cu:25-72, multi-GPU src/triangle/multigpu_base.cu:25-105
//   SglSolver     src/sgl/gpu_base.cu:21-103
//   CliqueSolver  src/clique/gpu_base.cu:14-80,  multi-GPU src/clique/multigpu.cu:20-139
//   MotifSolver   src/motif/gpu_base.cu:21-110
//
// Single GPU: upload (GraphGPU::init), one gm_* call, print the reference's runtime / throughput lines.
// Multi GPU (n_gpu > 1), one process, n devices, ONE HOST THREAD PER DEVICE (the reference's shape: src/triangle/multigpu.cu:67,
// src/clique/multigpu.cu:109 start a std::thread per GPU) -- a gm_* launch is several kernels + memsets + share-table lookups, and at
// 0.4 ms of kernel per rank eight of them issued one after the other by one thread were on the critical path:
//   * the CSR goes over PCIe ONCE (to GPU 0) and is replicated with ncclBroadcast over xGMI, instead of
//     the reference's n host->device copies (src/clique/multigpu.cu:57-66);
//   * the task split is index arithmetic on chunk ids inside the library (rank i owns chunks i mod n,
//     Scheduler::round_robin policy, src/common/scheduler.cc:34-85) -- no per-GPU COO copies;
//   * the per-GPU 64-bit counts are combined by ONE ncclAllReduce(ncclUint64, ncclSum) instead of the
//     host-side `total += h_counts[i]` (src/clique/multigpu.cu:134) / MPI_Allreduce (src/triangle/dist_cpu.cpp:56).
#include "multi.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/time.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <mutex>
#include <thread>
#include <vector>

namespace gmhost {
namespace {

struct Stopwatch {  // (the reference's Timer, include/timer.h:6-33: gettimeofday around the region)
  void Start() { gettimeofday(&a, nullptr); }
  void Stop() { gettimeofday(&b, nullptr); }
  double Seconds() const { return (b.tv_sec - a.tv_sec) + 1e-6 * (b.tv_usec - a.tv_usec); }
  timeval a{}, b{};
};

// n persistent host threads, thread i bound to device i: run(f) has every thread execute f(i) and returns when all have finished.
// (persistent: a std::thread costs ~50 us to start, which would sit inside every timed step)
class DeviceThreads {
 public:
  explicit DeviceThreads(int n) : n_(n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
  }
  ~DeviceThreads() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  void run(const std::function<void(int)> &f) {
    std::unique_lock<std::mutex> lk(mu_);
    job_ = &f;
    left_ = n_;
    ++gen_;
    cv_.notify_all();
    done_.wait(lk, [this] { return left_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int i) {
    (void)hipSetDevice(i);  // (the current device is per host thread)
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int)> *f = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
        f = job_;
      }
      (*f)(i);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--left_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *job_ = nullptr;
  unsigned long long gen_ = 0;
  int left_ = 0;
  bool quit_ = false;
};

// developer options of the library (include/graphminer_amd.h gm_dev_option): nothing here reads the environment
bool opt_set(const char *name) { return gm_dev_option_get(name) != nullptr; }

[[noreturn]] void die(int status, const char *where) {  // message + exit: CUDA_SAFE_CALL's reaction (include/cutil_subset.h:4-10)
  std::fprintf(stderr, "error: %s: %s", where, gm_strerror(status));
  const char *d = gm_last_error();
  if (d && *d) std::fprintf(stderr, " [%s]", d);
  std::fprintf(stderr, "\n");
  std::exit(EXIT_FAILURE);
}


#define HIP_OK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
      std::exit(EXIT_FAILURE);                                                                         \
    }                                                                                                  \
  } while (0)
#define NCCL_OK(call)                                                                                  \
  do {                                                                                                 \
    ncclResult_t r_ = (call);                                                                          \
    if (r_ != ncclSuccess) {                                                                           \
      std::fprintf(stderr, "RCCL error %s at %s:%d\n", ncclGetErrorString(r_), __FILE__, __LINE__);    \
      std::exit(EXIT_FAILURE);                                                                         \
    }                                                                                                  \
  } while (0)


bool sgl4(const Job &j) {
  return j.kind == Job::SGL && j.pattern &&
         (!std::strcmp(j.pattern, "tailedtriangle") || !std::strcmp(j.pattern, "4path") || !std::strcmp(j.pattern, "3star"));
}

int call(const Job &j, gm_graph *g, const gm_launch *la, uint64_t *out, gm_stats *st) {
  switch (j.kind) {
    case Job::TC: return gm_tc(g, la, out, st);
    case Job::SGL:
      // tailedtriangle / 4path / 3star on several GPUs: the four raw per-edge sums stay on the device, the all-reduce adds them,
      // run_multi applies gm_sgl4_finish (the halving / division by six needs the whole graph's sums)
      if (sgl4(j) && la && la->d_counts && !out) return gm_sgl4_partial(g, la, out, st);
      return gm_sgl(g, j.pattern, la, out, st);
    case Job::CLIQUE: return gm_clique(g, j.k, la, out, st);
    case Job::MOTIF:
      // 4-motif on several GPUs: every rank leaves its six RAW sums on the device (gm_motif4_partial), the all-reduce adds
      // them, run_multi applies gm_motif4_finish to the reduced sums (src/motif/omp_formula.cc:41-45)
      // (also with one device under the developer option GM_FORCE_RCCL_PATH, so that a one-GPU test box exercises this path)
      if (j.k == 4 && la && la->d_counts && !out) return gm_motif4_partial(g, la, out, st);
      return gm_motif(g, j.k, la, out, j.ncounts, st);
  }
  return GM_ERR_INVALID;
}

// returns false when the pattern / k is not implemented (caller prints the reference's message)
bool run_single(const gm_csr &h, const Job &j, int chunk, uint64_t *out) {
  gm_graph *dg = nullptr;
  int rc = gm_graph_upload(&h, 0, &dg);
  if (rc) die(rc, "gm_graph_upload");
  gm_launch la;
  std::memset(&la, 0, sizeof la);
  la.chunk = chunk > 0 ? chunk : 0;  // honoured as given (src/triangle/main.cc:16); <= 0 = the library default (the apps pass 0 when argv has none)
  gm_stats st;
  std::memset(&st, 0, sizeof st);
  rc = call(j, dg, &la, out, &st);  // first call builds the task-chunk table ("Time on generating the edgelist")
  if (rc == GM_ERR_UNSUPPORTED) { gm_graph_free(dg); return false; }
  if (rc) die(rc, "mining kernel");
  rc = call(j, dg, &la, out, &st);  // timed call: kernel only, like the reference's Timer (gpu_base.cu:53-65)
  if (rc) die(rc, "mining kernel");
  std::cout << "HIP " << j.name << " (" << st.grid << " workgroups, " << st.block << " threads/workgroup)\n";
  const double sec = st.kernel_ms * 1e-3;
  std::cout << "runtime [" << j.name << "] = " << sec << " sec\n";
  std::cout << "throughput = " << double(st.tasks) / sec / 1e9 << " billion Traversed Edges Per Second (TEPS)\n";
  gm_graph_free(dg);
  return true;
}

bool run_multi(const gm_csr &g, const Job &j, int n, int chunk, uint64_t *out) {
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = i;
  std::vector<ncclComm_t> comms(n);
  NCCL_OK(ncclCommInitAll(comms.data(), n, devs.data()));
  const size_t nv = size_t(g.nv), ne = size_t(g.ne);
  std::vector<int64_t *> d_rp(n);
  std::vector<int32_t *> d_ci(n);
  std::vector<uint64_t *> d_cnt(n);
  std::vector<hipStream_t> streams(n);
  Stopwatch tb;
  tb.Start();
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipSetDevice(i));
    HIP_OK(hipStreamCreate(&streams[i]));
    HIP_OK(hipMalloc(&d_rp[i], sizeof(int64_t) * (nv + 1)));
    HIP_OK(hipMalloc(&d_ci[i], sizeof(int32_t) * (ne ? ne : 1)));
    HIP_OK(hipMalloc(&d_cnt[i], sizeof(uint64_t) * 8));
  }
  HIP_OK(hipSetDevice(0));
  HIP_OK(hipMemcpyAsync(d_rp[0], g.row_ptr, sizeof(int64_t) * (nv + 1), hipMemcpyHostToDevice, streams[0]));
  HIP_OK(hipMemcpyAsync(d_ci[0], g.col_idx, sizeof(int32_t) * ne, hipMemcpyHostToDevice, streams[0]));
  NCCL_OK(ncclGroupStart());
  for (int i = 0; i < n; ++i) {
    NCCL_OK(ncclBroadcast(d_rp[i], d_rp[i], nv + 1, ncclInt64, 0, comms[i], streams[i]));
    NCCL_OK(ncclBroadcast(d_ci[i], d_ci[i], ne, ncclInt32, 0, comms[i], streams[i]));
  }
  NCCL_OK(ncclGroupEnd());
  std::vector<gm_graph *> dg(n, nullptr);
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipSetDevice(i));
    HIP_OK(hipStreamSynchronize(streams[i]));
    int rc = gm_graph_from_device(int32_t(nv), int64_t(ne), d_rp[i], d_ci[i], i, &dg[i]);
    if (rc) die(rc, "gm_graph_from_device");
  }
  tb.Stop();
  std::cout << "Time on replicating the CSR to " << n << " GPUs (1 PCIe copy + RCCL broadcast over xGMI): " << tb.Seconds()
            << " sec\n";

  // diamond on several GPUs: the one-GPU algorithm at every N (include/graphminer_amd.h gm_diamond_support_*): every GPU's share of the
  // triangle pass into its own support array, ONE ncclReduceScatter (uint32, sum) over xGMI, sum C(t, 2) of the received slice, then the
  // all-reduce of the counts like every other pattern.  (GM_DIAMOND_PER_EDGE, or rows beyond the 2048-entry stage: the per-edge kernels.)
  std::vector<uint32_t *> d_sup(n, nullptr);
  int64_t sup_n = 0;
  // Up to four GPUs: beyond, a GPU's share of the per-edge kernels + the single 8-byte all-reduce is the better deal (one-GPU simulation of
  // the shares on R-MAT-22 at 2 / 4 / 8 ranks: shared triangle pass 4.48 / 3.04 / 2.35 ms + a reduce-scatter of 81 / 122 / 142 MB per rank,
  // per-edge kernels 8.20 / 4.75 / 2.90 ms: profiles/r05/sim_scale_one_gpu.txt).  The developer option GM_DIAMOND_SUPPORTS_MAX_WORLD overrides the limit.
  int sup_max_world = 4;
  if (const char *e = gm_dev_option_get("GM_DIAMOND_SUPPORTS_MAX_WORLD")) sup_max_world = std::atoi(e);
  bool diamond_sup = j.kind == Job::SGL && j.pattern && std::strcmp(j.pattern, "diamond") == 0 && !opt_set("GM_DIAMOND_PER_EDGE") &&
                     (n <= sup_max_world || opt_set("GM_FORCE_RCCL_PATH"));
  if (diamond_sup) {
    for (int i = 0; i < n && diamond_sup; ++i) {
      HIP_OK(hipSetDevice(i));
      int64_t m = 0;
      int rc = gm_diamond_support_size(dg[i], n, &m);
      if (rc == GM_ERR_UNSUPPORTED) { diamond_sup = false; break; }
      if (rc) die(rc, "gm_diamond_support_size");
      sup_n = m;
      HIP_OK(hipMalloc(&d_sup[i], sizeof(uint32_t) * size_t(m)));
    }
  }
  DeviceThreads threads(n);  // thread i drives device i: its launches, its collectives, its synchronisation
  std::vector<gm_launch> las(n);
  for (int i = 0; i < n; ++i) {
    gm_launch &la = las[i];
    std::memset(&la, 0, sizeof la);
    la.stream = streams[i];
    la.rank = i;
    la.world = n;
    la.policy = GM_PART_ROUND_ROBIN;
    la.chunk = chunk > 0 ? chunk : 0;
    la.d_counts = d_cnt[i];
  }
  // (RCCL: one communicator per device, each used by its own thread -- collectives are enqueued on the device's stream without a group
  // call, which would have to be opened and closed by one thread for all of them)
  auto launch_diamond = [&]() {
    const size_t per = size_t(sup_n) / size_t(n);
    threads.run([&](int i) {
      int rc = gm_diamond_support_partial(dg[i], &las[i], d_sup[i], sup_n, nullptr);
      if (rc) die(rc, "gm_diamond_support_partial");
      // in place: GPU i receives its slice where it lies in its own array
      NCCL_OK(ncclReduceScatter(d_sup[i], d_sup[i] + size_t(i) * per, per, ncclUint32, ncclSum, comms[i], streams[i]));
      rc = gm_diamond_support_finish(dg[i], &las[i], d_sup[i] + size_t(i) * per, int64_t(per), nullptr, nullptr);
      if (rc) die(rc, "gm_diamond_support_finish");
      NCCL_OK(ncclAllReduce(d_cnt[i], d_cnt[i], 1, ncclUint64, ncclSum, comms[i], streams[i]));
      HIP_OK(hipStreamSynchronize(streams[i]));
    });
  };
  auto launch_all = [&](bool &unsupported) {
    if (diamond_sup) return launch_diamond();
    std::vector<int> rcs(n, GM_OK);
    threads.run([&](int i) {
      rcs[i] = call(j, dg[i], &las[i], nullptr, nullptr);  // asynchronous: returns when the kernels are enqueued
      // (every rank gets the same verdict on "unsupported" / invalid -- it depends on the pattern, not on the share -- so either all
      // ranks join the all-reduce or none does)
      if (rcs[i] != GM_OK) return;
      NCCL_OK(ncclAllReduce(d_cnt[i], d_cnt[i], sgl4(j) ? size_t(4) : size_t(j.ncounts), ncclUint64, ncclSum, comms[i], streams[i]));
      HIP_OK(hipStreamSynchronize(streams[i]));
    });
    for (int i = 0; i < n; ++i) {
      if (rcs[i] == GM_ERR_UNSUPPORTED) { unsupported = true; return; }
      if (rcs[i]) die(rcs[i], "mining kernel");
    }
  };
  bool unsupported = false;
  launch_all(unsupported);  // warm-up: builds the chunk tables
  if (!unsupported) {
    Stopwatch t;
    t.Start();
    launch_all(unsupported);
    t.Stop();
    for (int i = 0; i < n; ++i) {
      double ms = 0, two[2] = {0, 0};
      int got = 0;
      if (diamond_sup) {  // (two launches per step: the share of the triangle pass + sum C(t, 2) of the slice)
        gm_kernel_times(dg[i], 2, two, &got);
        ms = two[0] + two[1];
      } else {
        gm_kernel_times(dg[i], 1, &ms, &got);
      }
      std::cout << "runtime[gpu" << i << "] = " << ms * 1e-3 << " sec\n";  // src/clique/multigpu.cu:136-137
    }
    std::cout << "runtime [" << j.name << "] = " << t.Seconds() << " sec\n";
    HIP_OK(hipSetDevice(0));
    if (sgl4(j)) {  // reduced raw sums -> the pattern's count
      uint64_t raw[4];
      HIP_OK(hipMemcpy(raw, d_cnt[0], sizeof raw, hipMemcpyDeviceToHost));
      int rc = gm_sgl4_finish(j.pattern, raw, out);
      if (rc) die(rc, "gm_sgl4_finish");
    } else
    HIP_OK(hipMemcpy(out, d_cnt[0], sizeof(uint64_t) * size_t(j.ncounts), hipMemcpyDeviceToHost));
    if (j.kind == Job::MOTIF && j.k == 4) {  // reduced raw sums -> the six vertex-induced counts
      uint64_t raw[6];
      std::memcpy(raw, out, sizeof raw);
      int rc = gm_motif4_finish(raw, out);
      if (rc) die(rc, "gm_motif4_finish");
    }
  }
  for (int i = 0; i < n; ++i) {
    HIP_OK(hipSetDevice(i));
    gm_graph_free(dg[i]);
    if (d_sup[i]) HIP_OK(hipFree(d_sup[i]));
    HIP_OK(hipFree(d_rp[i]));
    HIP_OK(hipFree(d_ci[i]));
    HIP_OK(hipFree(d_cnt[i]));
    HIP_OK(hipStreamDestroy(streams[i]));
    ncclCommDestroy(comms[i]);
  }
  return !unsupported;
}

}  // namespace

bool run(const gm_csr &g, Job j, int n_gpu, int chunk, uint64_t *out) {
  int ndev = 0;
  int rc = gm_device_count(&ndev);
  if (rc) die(rc, "gm_device_count");
  if (n_gpu > ndev) {
    std::cout << "requested " << n_gpu << " GPUs, " << ndev << " available\n";
    n_gpu = ndev;
  }
  // gm_dev_option("GM_FORCE_RCCL_PATH", "1") drives the multi-GPU code (broadcast + all-reduce) even with one device,
  // so the RCCL path is exercised on a single-GPU test box.
  if (n_gpu <= 1 && !opt_set("GM_FORCE_RCCL_PATH")) return run_single(g, j, chunk, out);
  if (n_gpu < 1) n_gpu = 1;
  j.name = "multigpu";
  return run_multi(g, j, n_gpu, chunk, out);
}

}  // namespace gmhost
