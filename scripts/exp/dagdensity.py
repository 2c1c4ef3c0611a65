#!/usr/bin/env python3
"""How dense are the induced DAGs of the wide vertices? triangles (= pairs of the 4-clique count phase), streamed keys, by out-degree class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphminer_amd import TCSolver, CliqueSolver
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]), int(sys.argv[2])
sym, rp, ci = rmat_csr_device(scale, ef, 42, 0)
dag = sym.orient()
h = dag.download()
rp = torch.from_numpy(h.row_ptr).cuda(); ci = torch.from_numpy(h.col_idx).cuda().long()
nv = rp.numel() - 1
deg = rp[1:] - rp[:-1]
src = torch.repeat_interleave(torch.arange(nv, device="cuda"), deg)
print("DAG edges", ci.numel(), "TC", TCSolver(dag), "4-clique", CliqueSolver(dag, 4))
for lo, hi in ((0, 256), (256, 512), (512, 1024), (1024, 4096)):
    m = (deg[src] > lo) & (deg[src] <= hi)
    e = int(m.sum()); keys = int(deg[ci[m]].sum()); a = int(deg[src[m]].sum())
    print(f"d+ in ({lo},{hi}]: edges {e}, streamed keys sum d+(v) {keys} (mean list {keys/max(e,1):.1f}), sum d+(u) {a}")
