#!/usr/bin/env python3
"""TC on a uniform random graph (short lists, LiveJournal-like mean oriented degree ~9): the regime where the
flattened short-list path and the per-batch overhead dominate."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphminer_amd import TCSolver, MotifSolver, SglSolver, CliqueSolver
from graphminer_amd.graph import DeviceGraph
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4_800_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 43_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
s = torch.randint(0, nv, (m,), device=dev, generator=g)
d = torch.randint(0, nv, (m,), device=dev, generator=g)
keep = s != d
s, d = s[keep], d[keep]
keys = torch.unique(torch.cat([(s << 32) | d, (d << 32) | s]))
src = keys >> 32
col = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
rp = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
torch.cumsum(torch.bincount(src, minlength=nv), 0, out=rp[1:])
sym = DeviceGraph.from_device_ptrs(nv, int(col.numel()), rp.data_ptr(), col.data_ptr(), 0, keepalive=(rp, col))
dag = sym.orient()
print("nv", nv, "sym ne", sym.E(), "dag ne", dag.E(), "dag maxdeg", dag.get_max_degree())
for tune in ([0]*8, [256,1,0,0,0,0,0,0], [1024,1,0,0,0,0,0,0], [1024,1,0,0,0,0,4,0], [1024,1,0,0,0,1,0,0]):
    ms = []
    for _ in range(3):
        r, st = TCSolver(dag, tune=tune, return_stats=True); ms.append(st.kernel_ms)
    print("tc tune", tune, "kernel_ms", min(ms), "Medges/s", dag.E() / min(ms) / 1e3, "count", r)
for name, fn in (("diamond", lambda: SglSolver(sym, "diamond", return_stats=True)), ("motif3", lambda: MotifSolver(sym, 3, return_stats=True)),
                 ("clique4", lambda: CliqueSolver(dag, 4, return_stats=True))):
    r, st = fn(); r, st = fn()
    print(name, "kernel_ms", st.kernel_ms, "Medges/s", st.tasks / st.kernel_ms / 1e3, "count", r)
