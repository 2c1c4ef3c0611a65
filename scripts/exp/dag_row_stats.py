#!/usr/bin/env python3
"""TC / clique streaming by DAG host-row length: per range of d+(u), task edges and the keys of the partner lists N+(v)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]), int(sys.argv[2])
sym, rp, col = rmat_csr_device(scale, ef, 42)
dag = sym.orient()
h = dag.download()
rp = torch.from_numpy(h.row_ptr.astype('int64')).cuda(); col = torch.from_numpy(h.col_idx.astype('int64')).cuda()
nv = rp.numel() - 1
deg = rp[1:] - rp[:-1]
row_of = torch.repeat_interleave(torch.arange(nv, device=col.device), deg)
a = deg[row_of]; b = deg[col]
tot = int(b.sum()); mn = int(torch.minimum(a, b).sum())
print("DAG nv", nv, "edges", col.numel(), "max d+", int(deg.max()), "sum d+(v)", tot, "sum min(d+(u),d+(v))", mn)
edges = [0, 16, 64, 128, 256, 512, 1024, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (a > lo) & (a <= hi)
    print(f"d+(u) in ({lo:5d},{hi:10d}]: {int(((deg > lo) & (deg <= hi)).sum()):9d} rows, {int(m.sum()):11d} edges, keys d+(v) {int(b[m].sum()):13d} ({100.0 * int(b[m].sum()) / tot:5.1f} %), mean list {int(b[m].sum()) / max(int(m.sum()), 1):7.1f}")
