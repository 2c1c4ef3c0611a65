#!/usr/bin/env python3
"""SgL rectangle / house on one R-MAT graph with the heavy centres' counter maps in LDS (default) and in global memory (tune[6] & 0x20000,
round 5's form) on ONE handle: counts and kernel ms of both.  usage: lds_maps_check.py <scale> <edge factor> <pattern> [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphminer_amd.rmat import rmat_csr_device  # noqa: E402
from graphminer_amd.solvers import SglSolver  # noqa: E402


def main():
    sc, ef = int(sys.argv[1]), int(sys.argv[2])
    sym, _rp, _ci = rmat_csr_device(sc, ef, 42, 0)
    for pat in sys.argv[3:]:
        a = SglSolver(sym, pat)  # (first call: tables)
        a2, st = SglSolver(sym, pat, return_stats=True)
        print(pat, "maps in LDS", a2, f"{st.kernel_ms:.2f} ms", flush=True)
        b, stb = SglSolver(sym, pat, return_stats=True, tune=[0, 0, 0, 0, 0, 0, 0x20000])
        print(pat, "maps in global memory", b, f"{stb.kernel_ms:.2f} ms", "equal" if a == a2 == b else "DIFFERENT", flush=True)


if __name__ == "__main__":
    main()
