#!/usr/bin/env python3
"""Wall time of the 4-motif path (per-edge sums kernel + rectangle map kernel + 4-clique kernel + host fix-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphminer_amd import MotifSolver
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 18, int(sys.argv[2]) if len(sys.argv) > 2 else 16
sym, _rp, _ci = rmat_csr_device(scale, ef, 42, 0)
MotifSolver(sym, 4)
torch.cuda.synchronize()
t = time.perf_counter()
c = MotifSolver(sym, 4)
torch.cuda.synchronize()
print(f"rmat{scale}_ef{ef} 4-motif {1e3 * (time.perf_counter() - t):.2f} ms counts {c}")
