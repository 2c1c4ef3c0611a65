#!/usr/bin/env python3
"""A hub of 300 K neighbours (13 pieces of the giant-row kernel) + 2 hubs of 60 K + random edges: classes vs the general path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import oracle as O
from graphminer_amd import SglSolver, MotifSolver
from graphminer_amd.rmat import csr_from_pairs
rng = np.random.default_rng(11)
nv = 400000
s, d = [], []
for hub, n in ((7, 300000), (123456, 60000), (399999, 60000)):
    x = rng.choice(nv, n, replace=False).astype(np.uint64)
    s.append(np.full(n, hub, dtype=np.uint64)); d.append(x)
s.append(rng.integers(0, nv, 3000000).astype(np.uint64)); d.append(rng.integers(0, nv, 3000000).astype(np.uint64))
for v in range(1000, 1040):
    y = rng.choice(nv, 5000, replace=False).astype(np.uint64)
    s.append(np.full(y.size, v, dtype=np.uint64)); d.append(y)
g = csr_from_pairs(nv, np.concatenate(s), np.concatenate(d))
print("nv", nv, "entries", g.col_idx.size, "max degree", int(np.diff(g.row_ptr).max()))
osym = O.OGraph(g.row_ptr, g.col_idx)
want_d, want_m = O.diamond(osym), O.motif3(osym)
sd = g.to_device(0)
for name, tune in (("classes", None), ("general", [0, 0, 0, 0, 0, 0, 0x80000]), ("sorted classes + SPLIT giants", [0, 0, 0, 0, 0, 0, 0x400000 | 0x1000000])):
    gd, st = SglSolver(sd, "diamond", tune=tune, return_stats=True)
    gm, sm = MotifSolver(sd, 3, tune=tune, return_stats=True)
    print(f"{name:32s} diamond {gd == want_d} {st.kernel_ms:.2f} ms  motif3 {gm == want_m} {sm.kernel_ms:.2f} ms")
    assert gd == want_d and gm == want_m
parts = [SglSolver(sd, "diamond", rank=r, world=8) for r in range(8)]
assert sum(parts) == want_d
print("8 rank shares ok")
