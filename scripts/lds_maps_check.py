#!/usr/bin/env python3
"""SgL rectangle / house on one R-MAT graph with the heavy centres' counter maps in LDS (default) and in global memory (tune[6] & 0x20000,
round 5's form) on ONE handle: counts and kernel ms of both.
usage: lds_maps_check.py <scale> <edge factor> <pattern> [...]            R-MAT
       lds_maps_check.py uniform:<nv>,<m> 0 <pattern> [...]               flat degrees (LiveJournal-sized: uniform:4847571,43000000)
       lds_maps_check.py powerlaw:<nv>,<m>,<max degree> 0 <pattern> [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphminer_amd.rmat import powerlaw_csr_device, rmat_csr_device, uniform_csr_device  # noqa: E402
from graphminer_amd.solvers import SglSolver  # noqa: E402


def main():
    if sys.argv[1].startswith("uniform:"):
        nv, m = (int(x) for x in sys.argv[1].split(":")[1].split(","))
        sym, _rp, _ci = uniform_csr_device(nv, m)
    elif sys.argv[1].startswith("powerlaw:"):
        nv, m, md = (int(x) for x in sys.argv[1].split(":")[1].split(","))
        sym, _rp, _ci = powerlaw_csr_device(nv, m, md)
    else:
        sym, _rp, _ci = rmat_csr_device(int(sys.argv[1]), int(sys.argv[2]), 42, 0)
    for pat in sys.argv[3:]:
        a = SglSolver(sym, pat)  # (first call: tables)
        a2, st = SglSolver(sym, pat, return_stats=True)
        print(pat, "maps in LDS", a2, f"{st.kernel_ms:.2f} ms", flush=True)
        b, stb = SglSolver(sym, pat, return_stats=True, tune=[0, 0, 0, 0, 0, 0, 0x20000])
        print(pat, "maps in global memory", b, f"{stb.kernel_ms:.2f} ms", "equal" if a == a2 == b else "DIFFERENT", flush=True)


if __name__ == "__main__":
    main()
