import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle as O
from graphminer_amd import SglSolver, TCSolver
from graphminer_amd.rmat import powerlaw_csr_device
g, _rp, _ci = powerlaw_csr_device(4847571, 43000000, 20000, 2.5, 42, 0)
t=time.time(); a = SglSolver(g, "diamond"); b = SglSolver(g, "diamond", tune=[0,0,0,0,0,0,0x10000000]); c = SglSolver(g, "diamond", tune=[0,0,0,0,0,0,0x200])
d = SglSolver(g, "diamond", tune=[0,0,0,0,0,0,0x80000]); e = SglSolver(g, "diamond", tune=[0,0,0,0,0,0,0x800000])
print('support', a, 'per-edge', b, 'support as numbered', c, 'general kernel', d, 'support fallback', e, flush=True)
h = g.download()
t=time.time(); w = O.diamond(O.OGraph(h.row_ptr, h.col_idx)); print('oracle', w, time.time()-t)
