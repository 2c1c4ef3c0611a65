#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric ("million edges processed/sec + total match count") on MI355X.

A "step" = one pass of the hot path (one mining kernel over every task edge of the graph, plus the
RCCL all-reduce of the 64-bit count when --gpus > 1), inputs resident in HBM before the timed region.

Default workload (N=1) = BASELINE.json configs[1]: triangle counting on LiveJournal. The real
LiveJournal files are not in the image, so the default graph is the stand-in
SURVEY.md section 8d names: R-MAT scale 22, edge factor 10, seed 42 (|V| = 4.19 M, ~LiveJournal's
|E|); pass --graph <prefix> to run the real graph.meta.txt/vertex.bin/edge.bin instead.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

Multi-GPU: one process per GPU; every rank holds the full CSR (it regenerates / reloads it, no
broadcast needed), owns the task chunks c = rank (mod world) [Scheduler::round_robin policy,
src/common/scheduler.cc:34-85] and the per-rank counts are summed by ONE all-reduce per step
(torch.distributed backend "nccl" = RCCL over xGMI). Per-GPU work shrinks as N grows: scaling =
"strong" (the graph, hence total work, is fixed).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (default scale, edge factor, oriented?, description)
    "tc": (22, 10, True, "triangle counting, LiveJournal stand-in"),
    "diamond": (22, 10, False, "sgl diamond, LiveJournal stand-in"),
    "clique4": (22, 28, True, "4-clique, com-Orkut stand-in"),
    "motif3": (24, 16, False, "3-motif, R-MAT scale 24"),
    "rectangle": (16, 16, False, "sgl rectangle (4-cycle), R-MAT scale 16"),
    "house": (16, 16, False, "sgl house, R-MAT scale 16"),
    "pentagon": (16, 16, False, "sgl pentagon, R-MAT scale 16"),
    "clique5": (20, 16, True, "5-clique, R-MAT scale 20"),
    "motif3f": (24, 16, False, "3-motif, formula variant (motif_gpu_formula), R-MAT scale 24"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="tc", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=int, default=0)
    ap.add_argument("--ef", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", default="", help="prefix of graph.meta.txt/.vertex.bin/.edge.bin (real dataset)")
    ap.add_argument("--uniform", default="", help="NV,M: uniform random graph instead of R-MAT (LiveJournal-size flat-degree stand-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-baseline", action="store_true", help="skip the timed run of oracle/_ref/tc_omp_base")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the bounded oracle sample")
    ap.add_argument("--tune", default="", help="comma separated gm_launch.tune[] override")
    ap.add_argument("--policy", type=int, default=0, help="0 = chunked round robin, 1 = contiguous ranges")
    return ap.parse_args()


def alg_bytes(workload, rp, ci):
    """SURVEY.md 8(d) ALGORITHMIC bytes of one launch over the whole graph (numpy, exact)."""
    import numpy as np

    deg = np.diff(rp).astype(np.int64)
    sq = int((deg * deg).sum())          # sum_e d(src)
    dv = int(deg[ci].sum())              # sum_e d(dst)
    ne = int(ci.size)
    if workload in ("tc",):
        return 4 * (sq + dv) + 40 * ne
    if workload == "diamond":            # edges v1 < v0 only: by symmetry exactly half of sum_e (d(v0)+d(v1))
        return 4 * (sq + dv) // 2 + 40 * (ne // 2)
    if workload == "motif3":             # one difference per directed edge + one intersect per v1 < v0 edge
        return 4 * (sq + dv) + 4 * (sq + dv) // 2 + 40 * ne
    return None                          # clique4 needs |S1| per edge: taken from the oracle in tests, not here


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("GM_BENCH_FORCE_DIST") == "1"  # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from graphminer_amd import Graph, _lib
    from graphminer_amd._lib import gm_launch, gm_stats
    from graphminer_amd.rmat import rmat_csr_device

    lib = _lib.load()
    scale0, ef0, oriented, desc = WORKLOADS[a.workload]
    scale, ef = a.scale or scale0, a.ef or ef0

    # ---- input: CSR resident in HBM -------------------------------------------------------------
    t_in = time.perf_counter()
    if a.graph:
        sym = Graph(a.graph).to_device(local_rank)
        gname = f"file:{a.graph}"
    elif a.uniform:
        from graphminer_amd.rmat import uniform_csr_device

        unv, um = (int(x) for x in a.uniform.split(","))
        sym, _rp, _ci = uniform_csr_device(unv, um, a.seed, local_rank)
        gname = f"uniform_nv{unv}_m{um}_seed{a.seed}"
    else:
        sym, _rp, _ci = rmat_csr_device(scale, ef, a.seed, local_rank)
        gname = f"rmat_s{scale}_ef{ef}_seed{a.seed}"
    g = sym.orient() if oriented else sym
    torch.cuda.synchronize()
    t_in = time.perf_counter() - t_in

    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    la = gm_launch()
    la.stream = torch.cuda.current_stream().cuda_stream or None
    la.rank, la.world, la.policy = rank, world, a.policy
    la.d_counts = counts.data_ptr()
    if a.tune:
        for i, t in enumerate(a.tune.split(",")):
            la.tune[i] = int(t)
    st = gm_stats()

    def step():
        if a.workload == "tc":
            rc = lib.gm_tc(g.handle, C.byref(la), None, C.byref(st))
        elif a.workload == "diamond":
            rc = lib.gm_sgl(g.handle, b"diamond", C.byref(la), None, C.byref(st))
        elif a.workload in ("rectangle", "house", "pentagon"):
            rc = lib.gm_sgl(g.handle, a.workload.encode(), C.byref(la), None, C.byref(st))
        elif a.workload in ("clique4", "clique5"):
            rc = lib.gm_clique(g.handle, int(a.workload[-1]), C.byref(la), None, C.byref(st))
        elif a.workload == "motif3f":
            rc = lib.gm_motif_formula(g.handle, 3, C.byref(la), None, 2, C.byref(st))
        else:
            rc = lib.gm_motif(g.handle, 3, C.byref(la), None, 2, C.byref(st))
        _lib.check(rc, "bench step")
        if use_dist:
            dist.all_reduce(counts)  # ONE RCCL all-reduce of the 64-bit counts

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    result = [int(x) for x in counts.cpu().tolist()]
    kms = g.kernel_times_ms(min(a.steps, 64))
    k_avg_ms = sum(kms) / max(len(kms), 1)

    # "edges processed" = the reference's nnz (src/triangle/gpu_base.cu:69): |E+| (tc, clique),
    # ne/2 (diamond), ne (motif)
    tasks_total = g.E() // 2 if a.workload in ("diamond", "rectangle", "house", "pentagon") else g.E()
    ms_per_step = 1e3 * elapsed / a.steps
    value = tasks_total / (elapsed / a.steps) / 1e6

    out = {
        "metric": "million edges processed/sec + total match count",
        "value": round(value, 3),
        "unit": "Medges/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32 (vertex ids, offsets; uint64 counts)",
        "data": "file" if a.graph else ("synthetic uniform random graph (torch RNG)" if a.uniform else
                                        "synthetic R-MAT (0.57,0.19,0.19,0.05), SplitMix64 counter stream"),
        "config": {"workload": f"{a.workload}: {desc}", "graph": gname, "nv": g.V(), "ne_sym": sym.E(), "tasks": tasks_total,
                   "max_degree": g.get_max_degree(), "parallelism": f"task-chunk round-robin x{world}, replicated CSR",
                   "input_build_s": round(t_in, 2)},
        "count": result[:2] if a.workload.startswith("motif3") else result[0],
        "matches_per_sec": round((result[1] if a.workload.startswith("motif3") else result[0]) / (elapsed / a.steps), 1),
        "kernel_ms_avg": round(k_avg_ms, 4),
    }

    if rank == 0:
        host = g.download()
        ab = alg_bytes(a.workload, host.row_ptr, host.col_idx)
        if a.workload == "clique4":  # level 1 = the TC formula, level 2 from the statistics kernel
            l2 = C.c_uint64(0)
            _lib.check(lib.gm_clique4_level2_bytes(g.handle, C.byref(l2)), "gm_clique4_level2_bytes")
            ab = alg_bytes("tc", host.row_ptr, host.col_idx) + int(l2.value)
        if ab is not None:
            per_launch = ab / world  # each rank's kernel covers ~1/world of the chunks
            ach = per_launch / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):  # PMC-measured HBM bytes per launch (rocprofv3 --pmc), keyed by graph+workload
                traffic = json.load(open(tpath)).get(f"{a.workload}:{gname}:n{world}")
            out["roofline"] = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "algorithmic_bytes_per_launch": int(per_launch), "kernel": "gm::mine_kernel<PAT> (PAT = " + a.workload + ")"}
        if world == 1 and not a.no_cpu_baseline and a.workload == "tc":
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle as O  # the CPU oracle, timed as the OpenMP baseline ("port"): never the product path

            od = O.OGraph(host.row_ptr, host.col_idx)
            cal = 256
            t1 = time.perf_counter()
            O.tc_sample(od, cal, 1)
            tcal = time.perf_counter() - t1
            stride = max(1, int(tcal * cal / max(a.cpu_seconds, 1e-3)))
            t1 = time.perf_counter()
            cnt, tasks = O.tc_sample(od, stride, 0)
            tcpu = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(tasks / tcpu / 1e6, 3), "unit": "Medges/s", "cores": O.num_threads(),
                                   "kind": "port", "seconds": round(tcpu, 2),
                                   "sample": f"oracle gmo_tc_sample: vertices u = 0 mod {stride} of the same DAG "
                                             f"({tasks} of {g.E()} task edges), OpenMP schedule(dynamic,1)",
                                   "host_cpus": os.cpu_count()}
            if stride == 1:
                out["cpu_baseline"]["count_matches_gpu"] = bool(cnt == result[0])
            ref = None if a.no_ref_baseline else reference_tc_baseline(sym, g.E(), O.num_threads(), result[0])
            if ref is not None:  # the reference's own tc_omp_base binary (oracle/_ref, prebuilt) on the same graph
                ref["port"] = {k: out["cpu_baseline"][k] for k in ("value", "seconds", "sample")}
                out["cpu_baseline"] = ref
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()  # rank 0 finishes its host-side reporting before any rank tears the communicator down
        dist.destroy_process_group()


def reference_tc_baseline(sym, tasks, threads, gpu_count):
    """Time the REFERENCE's tc_omp_base (built by oracle/ref/Makefile into oracle/_ref/, test infrastructure) on the same
    graph: write the symmetric CSR in the three-file format, run the binary, read its own `runtime [omp_base]` line
    (Timer around the parallel loop only, src/triangle/omp_base.cc:12-24). None when the binary is not there or fails."""
    import re
    import shutil
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "tc_omp_base")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="gm_ref_", dir="/tmp")
    try:
        sym.download().save(os.path.join(tmp, "graph"))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread")
        best, count = None, None
        for _ in range(2):
            r = subprocess.run([exe, os.path.join(tmp, "graph")], capture_output=True, text=True, env=env, timeout=300)
            m = re.search(r"runtime \[omp_base\] = ([0-9.eE+-]+) sec", r.stdout)
            c = re.search(r"total_num_triangles = (\d+)", r.stdout)
            if r.returncode != 0 or not m or not c:
                return None
            t = float(m.group(1))
            best, count = (t if best is None else min(best, t)), int(c.group(1))
        return {"value": round(tasks / best / 1e6, 3), "unit": "Medges/s", "cores": threads, "kind": "reference",
                "seconds": round(best, 3), "count_matches_gpu": bool(count == gpu_count),
                "sample": "tc_omp_base (reference binary, g++ -O3 -fopenmp, its own orientation + Timer) on the whole graph, "
                          "best of 2 runs", "host_cpus": os.cpu_count()}
    except Exception as e:  # the baseline is a report, never a reason to lose the bench line
        print(f"[bench] reference baseline skipped: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
