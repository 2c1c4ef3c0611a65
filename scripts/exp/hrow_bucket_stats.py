#!/usr/bin/env python3
"""Bucket statistics of the hashed-row classes (gm_hrow.hip) on the device-generated R-MAT graph: per class, how many entries
overflow their 8-slot bucket, how many buckets are flagged, the largest surplus list of a row."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd.rmat import rmat_csr_device
scale, ef = int(sys.argv[1]), int(sys.argv[2])
g, rp, col = rmat_csr_device(scale, ef, 42)
nv = rp.numel() - 1
deg = rp[1:] - rp[:-1]
K = max((nv - 1).bit_length(), 1)
GOLD = 0x9E3779B97F4A7C15
ck = ((GOLD >> (64 - K)) | 1)
print("nv", nv, "K", K, "ck", hex(ck))
row_of = torch.repeat_interleave(torch.arange(nv, device=col.device), deg)
for name, lo, hi, lbmax in (("class1", 3072, 8191, 11), ("class2", 8191, 24576, 13)):
    sel_rows = (deg > lo) & (deg <= hi)
    m = sel_rows[row_of]
    keys = col[m].to(torch.int64); rows = row_of[m]
    n_row = deg[rows]
    lb = torch.clamp(torch.maximum(torch.tensor(K - 14, device=col.device), torch.floor(torch.log2(((n_row - 1) // 3).double())).long() + 1), max=min(lbmax, K))
    h = (keys * ck) & ((1 << K) - 1)
    b = h >> (K - lb)
    bid = rows * (1 << 13) + b
    uniq, cnt = torch.unique(bid, return_counts=True)
    over = torch.clamp(cnt - 8, min=0)
    per_row = torch.zeros(nv, dtype=torch.int64, device=col.device).index_add_(0, uniq >> 13, over)
    flagged = int((cnt > 8).sum()); nb = int((1 << lb[torch.unique(rows, return_inverse=False).new_zeros(1)]).sum()) if False else 0
    print(f"{name}: rows {int(sel_rows.sum())} entries {keys.numel()} overflow {int(over.sum())} ({100.0 * int(over.sum()) / max(keys.numel(), 1):.4f} %), flagged buckets {flagged}, "
          f"largest bucket {int(cnt.max())}, largest surplus of a row {int(per_row.max())}, rows with surplus > 16: {int((per_row > 16).sum())}, > 128: {int((per_row > 128).sum())}")
