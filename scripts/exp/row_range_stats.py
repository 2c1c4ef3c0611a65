#!/usr/bin/env python3
"""Where the streamed keys of the symmetric-graph patterns sit: per range of host-row length, rows / task edges / streamed keys
(every undirected edge is hosted by its longer row, which streams the shorter list)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]), int(sys.argv[2])
g, rp, col = rmat_csr_device(scale, ef, 42)
nv = rp.numel() - 1
deg = rp[1:] - rp[:-1]
row_of = torch.repeat_interleave(torch.arange(nv, device=col.device), deg)
a = deg[row_of]; b = deg[col.long()]
hosts = (a > b) | ((a == b) & (row_of > col))
keys = torch.where(hosts, b, torch.zeros_like(b))
edges = [0, 64, 256, 1024, 3072, 8191, 24576, 49152, 98304, 1 << 30]
tot = int(keys.sum())
print("nv", nv, "entries", col.numel(), "max degree", int(deg.max()), "streamed keys", tot)
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (a > lo) & (a <= hi)
    print(f"rows ({lo:6d},{hi:10d}]: {int(((deg > lo) & (deg <= hi)).sum()):9d} rows, {int((m & hosts).sum()):11d} task edges, keys {int(keys[m].sum()):14d} ({100.0 * int(keys[m].sum()) / tot:5.1f} %), mean list {int(keys[m].sum()) / max(int((m & hosts).sum()), 1):8.1f}")
