#!/usr/bin/env python3
"""Simulated strong scaling on ONE GPU: run every rank's task share (world = 1, 2, 4, 8) back to back and report
max / mean kernel time per world size (what an N-GPU run would wait for, all-reduce excluded)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from graphminer_amd import CliqueSolver, MotifSolver, SglSolver, TCSolver
from graphminer_amd.rmat import rmat_csr_device

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="tc")
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--ef", type=int, default=10)
ap.add_argument("--policy", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--tune", default="")
ap.add_argument("--worlds", default="1,2,4,8")
a = ap.parse_args()
sym, _rp, _ci = rmat_csr_device(a.scale, a.ef, 42, 0)
dag = sym.orient()
run = {
    "tc": lambda **kw: TCSolver(dag, return_stats=True, **kw),
    "diamond": lambda **kw: SglSolver(sym, "diamond", return_stats=True, **kw),
    "clique4": lambda **kw: CliqueSolver(dag, 4, return_stats=True, **kw),
    "motif3": lambda **kw: MotifSolver(sym, 3, return_stats=True, **kw),
}.get(a.workload)
def diamond_sup(rank=0, world=1, policy=0, chunk=0, tune=None):
    """a rank's part of the several-rank diamond (gm_diamond_support_partial + _finish on its slice; the reduce-scatter itself -- 4 B per
    DAG entry over xGMI -- is not simulated): kernel ms of the two launches"""
    from graphminer_amd.solvers import diamond_support_finish, diamond_support_partial, diamond_support_size
    n = diamond_support_size(sym, world)
    buf = torch.empty(n, dtype=torch.int32, device="cuda:0")
    st1 = diamond_support_partial(sym, buf.data_ptr(), n, rank=rank, world=world, policy=policy, chunk=chunk, tune=tune, return_stats=True)
    per = n // world
    _, st2 = diamond_support_finish(sym, buf[rank * per:(rank + 1) * per].data_ptr(), per, return_stats=True)
    st1.kernel_ms += st2.kernel_ms
    return None, st1


if a.workload == "diamond_sup":
    run = diamond_sup
base = None
tune = [int(x) for x in a.tune.split(',')] if a.tune else None
for world in [int(x) for x in a.worlds.split(',')]:
    times = []
    for r in range(world):
        best = 1e30
        for _ in range(a.reps):
            _, st = run(rank=r, world=world, policy=a.policy, chunk=a.chunk, tune=tune)
            best = min(best, st.kernel_ms)
        times.append(best)
    mx, mean = max(times), sum(times) / len(times)
    if base is None:
        base = mx
    print(f"{a.workload} world={world} max={mx:.3f} ms mean={mean:.3f} ms skew={mx/mean:.3f} speedup={base/mx:.2f} eff={base/mx/world:.3f}")
