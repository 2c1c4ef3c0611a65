"""Experiment: does degree-ordered relabelling change the TC kernel time? (count must not change)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from graphminer_amd import TCSolver, DeviceGraph
from graphminer_amd.rmat import rmat_csr_device

scale, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 22, int(sys.argv[2]) if len(sys.argv) > 2 else 10
sym, rp, ci = rmat_csr_device(scale, ef, 42, 0)
nv = rp.numel() - 1
def tc_time(s):
    d = s.orient()
    best = 1e9
    for _ in range(4):
        c, st = TCSolver(d, return_stats=True)
        best = min(best, st.kernel_ms)
    return c, best
print("original", tc_time(sym))
deg = rp[1:] - rp[:-1]
src = torch.repeat_interleave(torch.arange(nv, device=rp.device), deg)
for name, order in (("deg-desc", torch.argsort(deg, descending=True, stable=True)), ("deg-asc", torch.argsort(deg, stable=True)),
                    ("random", torch.randperm(nv, device=rp.device))):
    newid = torch.empty(nv, dtype=torch.int64, device=rp.device)
    newid[order] = torch.arange(nv, device=rp.device)
    keys = torch.sort((newid[src] << 32) | newid[ci.long()]).values
    s2 = keys >> 32
    c2 = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
    rp2 = torch.zeros(nv + 1, dtype=torch.int64, device=rp.device)
    torch.cumsum(torch.bincount(s2, minlength=nv), 0, out=rp2[1:])
    g2 = DeviceGraph.from_device_ptrs(nv, int(c2.numel()), rp2.data_ptr(), c2.data_ptr(), 0, keepalive=(rp2, c2))
    print(name, tc_time(g2))
