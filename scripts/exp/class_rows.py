#!/usr/bin/env python3
"""entries (task edges) in the rows of each workgroup class, for the graphs of the A/B runs"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd import rmat
kind = sys.argv[1]
if kind == "rmat": g, rp, col = rmat.rmat_csr_device(int(sys.argv[2]), int(sys.argv[3]), 42)
elif kind == "powerlaw": g, rp, col = rmat.powerlaw_csr_device(4847571, 43000000, 20000, seed=42)
deg = rp[1:] - rp[:-1]
out = []
for lo, hi in ((3072, 8191), (8191, 24576), (24576, 1 << 30)):
    m = (deg > lo) & (deg <= hi)
    out.append(f"({lo},{hi}]: {int(m.sum())} rows {int(deg[m].sum())} entries")
print(" ".join(sys.argv[1:]), "nv", deg.numel(), "entries", int(deg.sum()), "maxdeg", int(deg.max()), " | ".join(out))
