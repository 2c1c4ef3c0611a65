#!/usr/bin/env python3
"""Tuning sweep on one graph: kernel_ms for a list of gm_launch.tune settings (GPU box only)."""
import argparse, itertools, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphminer_amd import TCSolver, SglSolver, CliqueSolver, MotifSolver
from graphminer_amd.rmat import rmat_csr_device

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22); ap.add_argument("--ef", type=int, default=10)
ap.add_argument("--workload", default="tc"); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--T", default="512"); ap.add_argument("--grab", default="1"); ap.add_argument("--cx", default="0")
ap.add_argument("--cy", default="0"); ap.add_argument("--bpc", default="0"); ap.add_argument("--nostage", default="0"); ap.add_argument("--dbg", default="0"); ap.add_argument("--base", default="0")
a = ap.parse_args()
sym, rp, ci = rmat_csr_device(a.scale, a.ef, 42, 0)
g = sym.orient() if a.workload in ("tc", "clique4") else sym
fn = {"tc": lambda **k: TCSolver(g, **k), "diamond": lambda **k: SglSolver(g, "diamond", **k),
      "clique4": lambda **k: CliqueSolver(g, 4, **k), "motif3": lambda **k: MotifSolver(g, 3, **k)}[a.workload]
L = lambda s: [int(x) for x in s.split(",")]
ref_count = None
for T, gr, cx, cy, bpc, ns, dbg, base in itertools.product(L(a.T), L(a.grab), L(a.cx), L(a.cy), L(a.bpc), L(a.nostage), L(a.dbg), L(a.base)):
    tune = [T, gr, cx, cy, bpc, ns, dbg, base]
    ms = []
    for _ in range(a.reps):
        r, st = fn(tune=tune, return_stats=True)
        ms.append(st.kernel_ms)
    if ref_count is None: ref_count = r
    print(f"maxdeg={g.get_max_degree()} ne={g.E()} tune={tune} kernel_ms min={min(ms):.3f} med={sorted(ms)[len(ms)//2]:.3f} count_ok={r == ref_count} grid={st.grid}", flush=True)
