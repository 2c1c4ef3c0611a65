#!/usr/bin/env python3
"""Full-size parity evidence: GPU counts vs the CPU oracle (OpenMP, all host cores) on the BASELINE stand-in graphs.
Usage: fullsize_check.py <workload: diamond|motif3|clique4> <scale> <ef>   -- prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import oracle as O
from graphminer_amd import CliqueSolver, MotifSolver, SglSolver
from graphminer_amd.rmat import rmat_csr_device

w, scale, ef = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sym, _rp, _ci = rmat_csr_device(scale, ef, 42, 0)
host = sym.download()
og = O.OGraph(host.row_ptr, host.col_idx)
if w == "clique4":
    gpu, st = CliqueSolver(sym.orient(), 4, return_stats=True)
    og = O.orient(og)
else:
    gpu, st = (SglSolver(sym, "diamond", return_stats=True) if w == "diamond" else MotifSolver(sym, 3, return_stats=True))
t = time.perf_counter()
cpu = O.clique(og, 4) if w == "clique4" else (O.diamond(og) if w == "diamond" else O.motif3(og))
dt = time.perf_counter() - t
print(json.dumps({"workload": w, "graph": f"rmat_s{scale}_ef{ef}_seed42", "gpu": gpu, "oracle": cpu, "equal": gpu == cpu,
                  "gpu_kernel_ms": round(st.kernel_ms, 3), "oracle_seconds": round(dt, 2), "oracle_threads": O.num_threads()}))
