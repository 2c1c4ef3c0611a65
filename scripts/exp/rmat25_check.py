#!/usr/bin/env python3
"""R-MAT-25 (ef 16: nv = 2^25, ~1e9 CSR entries): the class kernels of id spaces beyond 2^24 against the general path, and the
identity sum_v C(d,2) - 3T = wedges between 3-motif and TC."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd import SglSolver, MotifSolver, TCSolver
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]), int(sys.argv[2])
t = time.time(); sym, rp, ci = rmat_csr_device(scale, ef, 42, 0); print("graph", ci.numel(), "entries", round(time.time() - t, 1), "s")
G = [0, 0, 0, 0, 0, 0, 0x80000]
d0, st0 = SglSolver(sym, "diamond", return_stats=True); d0, st0 = SglSolver(sym, "diamond", return_stats=True)
m0, sm0 = MotifSolver(sym, 3, return_stats=True); m0, sm0 = MotifSolver(sym, 3, return_stats=True)
print("classes: diamond", d0, round(st0.kernel_ms, 1), "ms; 3-motif", m0, round(sm0.kernel_ms, 1), "ms")
d1, st1 = SglSolver(sym, "diamond", tune=G, return_stats=True)
m1, sm1 = MotifSolver(sym, 3, tune=G, return_stats=True)
print("general: diamond", d1 == d0, round(st1.kernel_ms, 1), "ms; 3-motif", m1 == m0, round(sm1.kernel_ms, 1), "ms")
tc = TCSolver(sym.orient())
deg = (rp[1:] - rp[:-1])
wedges = int((deg * (deg - 1) // 2).sum().item()) - 3 * tc
print("identity: triangles", tc == m0[1], "wedges", wedges % 2**64 == m0[0])
assert d1 == d0 and m1 == m0 and tc == m0[1] and wedges % 2**64 == m0[0]
