#!/usr/bin/env python3
"""CPU model of the 4-clique level-1 gather (csrc/gm_cgather.hip) on the bench graph: rows / probes / 128-byte lines of the current
row-major gather, and the traffic of a BLOCKED (transposed) gather -- blocks of core rows resident in LDS, the column tables of the
vertices streamed past them -- for fixed and span-sized block heights.  numpy only; prints a table.  (round 6)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphminer_amd.rmat import rmat_csr_numpy

scale, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 22, int(sys.argv[2]) if len(sys.argv) > 2 else 28
H = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
cache = f"/tmp/an/dag_{scale}_{ef}.npz"
t0 = time.time()
if os.path.exists(cache):
    z = np.load(cache); rp, col = z["rp"], z["col"]
else:
    g = rmat_csr_numpy(scale, ef, 42)
    nv = len(g.row_ptr) - 1
    deg = np.diff(g.row_ptr)
    order = np.lexsort((np.arange(nv), deg))            # ascending (degree, id)
    newid = np.empty(nv, np.int64); newid[order] = np.arange(nv)
    src = np.repeat(np.arange(nv), deg)
    s2, d2 = newid[src], newid[g.col_idx]
    keep = d2 > s2                                        # DAG: towards the higher (degree, id)
    s2, d2 = s2[keep], d2[keep]
    k = np.sort((s2 << 32) | d2)
    s2, d2 = k >> 32, (k & 0xffffffff).astype(np.int32)
    rp = np.zeros(nv + 1, np.int64); np.cumsum(np.bincount(s2, minlength=nv), out=rp[1:])
    col = d2
    np.savez(cache, rp=rp, col=col)
nv = len(rp) - 1
dplus = np.diff(rp)
print(f"nv {nv} |E+| {len(col)} max d+ {dplus.max()}  ({time.time()-t0:.0f} s)")
base = nv - H
wide = np.nonzero((dplus > 256) & (dplus <= 2048))[0]
print(f"wide vertices {len(wide)}, their edges {dplus[wide].sum()}")
rows = probes = lines128 = lines64 = words = 0
LINE = 128 * 8
# blocked variants: name -> function(position p in [0,H)) -> block id
def fixed(R): return lambda p: p // R
def span_blocks(lds_bytes):
    # consecutive rows packed while sum of row spans (H - p bits, rounded to 4 bytes) fits the LDS budget
    bid = np.zeros(H, np.int64); b = 0; used = 0
    for p in range(H):
        need = ((H - p + 31) // 32) * 4
        if used + need > lds_bytes: b += 1; used = 0
        bid[p] = b; used += need
    return bid
variants = {"R16": np.arange(H) // 16, "R32": np.arange(H) // 32, "R64": np.arange(H) // 64,
            "span64K": span_blocks(65536), "span128K": span_blocks(131072), "span32K": span_blocks(32768)}
units = {k: 0 for k in variants}; tbytes2 = {k: 0 for k in variants}; tbytes_trim = {k: 0 for k in variants}
hist_rows = np.zeros(H // 1024, np.int64); hist_probes = np.zeros(H // 1024, np.int64); hist_lines = np.zeros(H // 1024, np.int64)
for u in wide:
    s = col[rp[u]:rp[u + 1]].astype(np.int64)
    d = len(s)
    pos = s - base
    k0 = np.searchsorted(pos, 0)
    pc = pos[k0:]                     # the core part, ascending
    n = len(pc)
    if n == 0: continue
    rows += n
    # row i (core index) probes the columns j > i
    pr = n - 1 - np.arange(n)
    probes += pr.sum()
    ln = np.unique(pc // LINE)        # lines of a core row touched by this vertex's columns: row i touches those of columns > i
    lid = pc // LINE
    # number of distinct lines among pc[i+1:]: distinct line ids are ascending; count = (#distinct lines) - (index of line of pc[i+1] among distinct)
    first_idx = np.searchsorted(ln, lid)            # for each column, index of its line
    nl = np.where(pr > 0, len(ln) - np.append(first_idx[1:], len(ln)), 0)
    lines128 += nl.sum()
    wd = np.unique(pc // 32); wid = np.searchsorted(wd, pc // 32)
    words += np.where(pr > 0, len(wd) - np.append(wid[1:], len(wd)), 0).sum()
    hb = np.minimum(pc // 1024, H // 1024 - 1)
    np.add.at(hist_rows, hb, 1); np.add.at(hist_probes, hb, pr); np.add.at(hist_lines, hb, nl)
    for name, bid in variants.items():
        b = bid[pc]
        ub, first = np.unique(b, return_index=True)
        units[name] += len(ub)
        tbytes2[name] += len(ub) * 2 * n                      # whole column table per (vertex, block), 2 bytes per column
        tbytes_trim[name] += (2 * (n - first)).sum()          # only the columns from the block's first row on
print(f"rows (u, i) {rows}  probes {probes}  dwords {words} ({words*4/1e9:.1f} GB)  128-B lines {lines128} ({lines128*128/1e9:.1f} GB)")
print("position (K cols from core base): rows M / probes G / lines GB")
for i in range(H // 1024):
    print(f"  {i:3d}: {hist_rows[i]/1e6:8.2f} {hist_probes[i]/1e9:8.2f} {hist_lines[i]*128/1e9:8.2f}")
for name in variants:
    print(f"{name:9s} blocks {variants[name].max()+1:5d} units {units[name]/1e6:7.2f} M  rows/unit {rows/units[name]:5.2f}  table bytes whole {tbytes2[name]/1e9:6.1f} GB  trimmed {tbytes_trim[name]/1e9:6.1f} GB")
