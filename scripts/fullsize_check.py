#!/usr/bin/env python3
"""Full-size parity evidence: GPU counts vs the CPU oracle (OpenMP, all host cores) on the BASELINE stand-in graphs.
Usage: fullsize_check.py <workload: diamond|motif3|motif3f|clique4> <scale> <ef> [--ref]   (motif3f: --ref runs motif_omp_formula)   -- prints one JSON line.
--ref: the CPU answer comes from the REFERENCE's own binary (oracle/_ref/{sgl,motif,clique}_omp_base, built by oracle/ref/Makefile from the
sources under /root/reference; all host threads, its own Timer) instead of the oracle restatement -- VERDICT r3 item 8: the full-size
answers of tests/golden/fullsize.json pinned to the reference once."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import oracle as O
from graphminer_amd import CliqueSolver, MotifSolver, SglSolver
from graphminer_amd.rmat import rmat_csr_device

w, scale, ef = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
use_ref = "--ref" in sys.argv[4:]
sym, _rp, _ci = rmat_csr_device(scale, ef, 42, 0)
host = sym.download()
og = O.OGraph(host.row_ptr, host.col_idx)
if w == "clique4":
    gpu, st = CliqueSolver(sym.orient(), 4, return_stats=True)
    og = O.orient(og)
else:
    gpu, st = (SglSolver(sym, "diamond", return_stats=True) if w == "diamond" else MotifSolver(sym, 3, return_stats=True))
if use_ref:
    import re, subprocess, tempfile, shutil
    tmp = tempfile.mkdtemp(prefix="gm_full_", dir="/tmp")
    try:
        host.save(os.path.join(tmp, "graph"))
        exe, args, pat, nvals = {"clique4": ("clique_omp_base", ["4"], r"num_4-cliques = (\d+)", 1), "diamond": ("sgl_omp_base", ["diamond"], r"total_num = (\d+)", 1),
                                 "motif3": ("motif_omp_base", ["3"], r"pattern \d+: (\d+)", 2),
                                 "motif3f": ("motif_omp_formula", ["3"], r"pattern \d+: (\d+)", 2)}[w]
        env = dict(os.environ, OMP_NUM_THREADS=str(O.num_threads()), OMP_PROC_BIND="spread")
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), os.path.join(tmp, "graph"), *args], capture_output=True, text=True, env=env, timeout=7200)
        c = [int(x) for x in re.findall(pat, r.stdout)]
        m = re.search(r"runtime(?: \[[a-z_]+\])? = ([0-9.eE+-]+)", r.stdout)
        assert r.returncode == 0 and len(c) >= nvals and m, (r.returncode, r.stdout[-500:], r.stderr[-500:])
        cpu = c[-1] if nvals == 1 else c[-nvals:]
        print(json.dumps({"workload": w, "graph": f"rmat_s{scale}_ef{ef}_seed42", "gpu": gpu, "reference": cpu, "equal": gpu == cpu, "binary": "oracle/_ref/" + exe,
                          "gpu_kernel_ms": round(st.kernel_ms, 3), "reference_seconds": float(m.group(1)), "threads": O.num_threads()}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0)
t = time.perf_counter()
cpu = O.clique(og, 4) if w == "clique4" else (O.diamond(og) if w == "diamond" else O.motif3(og))
dt = time.perf_counter() - t
print(json.dumps({"workload": w, "graph": f"rmat_s{scale}_ef{ef}_seed42", "gpu": gpu, "oracle": cpu, "equal": gpu == cpu,
                  "gpu_kernel_ms": round(st.kernel_ms, 3), "oracle_seconds": round(dt, 2), "oracle_threads": O.num_threads()}))
