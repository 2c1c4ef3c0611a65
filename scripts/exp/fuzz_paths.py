#!/usr/bin/env python3
"""Differential fuzz on the GPU: every pattern through its default path and through the alternative implementations behind the A/B
switches (TC: hashed set / its fallback lookup / sorted copy / chunked kernel; diamond: edge supports vs the per-edge kernels three ways;
3-motif: formula vs enumeration; 4-clique: re-hosted build vs the arena path), on random graphs of varied shape, with rank shares.
usage: fuzz_paths.py [cases] [oracle]   (oracle: also against the CPU oracle for graphs below 2.5 M entries)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from graphminer_amd import TCSolver, SglSolver, MotifSolver, CliqueSolver
from graphminer_amd.rmat import csr_from_pairs, rmat_csr_numpy
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
use_oracle = len(sys.argv) > 2 and sys.argv[2] == "oracle"  # also against the CPU oracle (graphs below 2.5 M entries)
if use_oracle:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    import oracle as O
rng = np.random.default_rng(int(os.environ.get("GM_FUZZ_SEED", "2026")))
bad = 0
t0 = time.time()
for case in range(n_cases):
    kind = case % 5
    if kind == 0:
        sc = int(rng.integers(10, 18)); ef = int(rng.integers(4, 33)); g = rmat_csr_numpy(sc, ef, int(rng.integers(1, 1000))); name = f"rmat{sc}_ef{ef}"
    elif kind == 1:  # hubs of random sizes over a random background
        nv = int(rng.integers(20000, 200000)); s, d = [rng.integers(0, nv, nv * 6).astype(np.uint64)], [rng.integers(0, nv, nv * 6).astype(np.uint64)]
        for _ in range(int(rng.integers(1, 6))):
            n = int(rng.choice([1500, 5000, 9000, 20000, 30000, 60000])); n = min(n, nv - 1)
            s.append(np.full(n, int(rng.integers(0, nv)), dtype=np.uint64)); d.append(rng.choice(nv, n, replace=False).astype(np.uint64))
        g = csr_from_pairs(nv, np.concatenate(s), np.concatenate(d)); name = f"hubs_nv{nv}"
    elif kind == 2:  # dense-ish small graph (wide DAG rows)
        nv = int(rng.integers(1500, 5000)); m = nv * int(rng.integers(100, 400))
        g = csr_from_pairs(nv, rng.integers(0, nv, m).astype(np.uint64), rng.integers(0, nv, m).astype(np.uint64)); name = f"dense_nv{nv}_m{m}"
    elif kind == 4:  # power law with many short DAG rows per chunk (what caught the remainder-only hashed set)
        from graphminer_amd.rmat import powerlaw_csr_device
        nvp = int(rng.integers(200000, 3000000)); mp = nvp * int(rng.integers(4, 12))
        symp, _a, _b = powerlaw_csr_device(nvp, mp, int(rng.choice([2000, 20000, 100000])), 2.5, int(rng.integers(1, 1000)), 0)
        g = symp.download(); symp.free(); name = f"powerlaw_nv{nvp}_m{mp}"
    else:  # flat degrees
        nv = int(rng.integers(50000, 400000)); m = nv * int(rng.integers(3, 12))
        g = csr_from_pairs(nv, rng.integers(0, nv, m).astype(np.uint64), rng.integers(0, nv, m).astype(np.uint64)); name = f"flat_nv{nv}_m{m}"
    sym = g.to_device(0); dag = sym.orient()
    T = lambda f: [0, 0, 0, 0, 0, 0, f]
    P = 0x10000000  # diamond / 3-motif: one intersection per edge of the symmetric graph instead of the DAG's triangles
    res = {
        # hashed set (default), hashed set on its fallback lookup, sorted copy + filter + bisection, chunked kernel, as numbered, shares
        "tc": [TCSolver(dag), TCSolver(dag, tune=T(0x800000)), TCSolver(dag, tune=T(0x8000000)), TCSolver(dag, tune=T(0x4000000)), TCSolver(dag, tune=T(0x200)),
               sum(TCSolver(dag, rank=r, world=3) for r in range(3)), CliqueSolver(dag, 3)],
        # edge supports (default), supports on the fallback lookup / as numbered / with eager parts, per-edge kernels three ways, shares
        "diamond": [SglSolver(sym, "diamond"), SglSolver(sym, "diamond", tune=T(0x800000)), SglSolver(sym, "diamond", tune=T(0x200)), SglSolver(sym, "diamond", tune=T(0x1000)),
                    SglSolver(sym, "diamond", tune=T(P)), SglSolver(sym, "diamond", tune=T(0x80000)), SglSolver(sym, "diamond", tune=T(0x100000 | 0x400000 | 0x1000000)),
                    sum(SglSolver(sym, "diamond", rank=r, world=5) for r in range(5))],
        "motif3": [MotifSolver(sym, 3), MotifSolver(sym, 3, tune=T(P)), MotifSolver(sym, 3, tune=T(0x80000)), MotifSolver(sym, 3, tune=T(0x100000 | 0x2000000)),
                   [sum(x) % 2**64 for x in zip(*[MotifSolver(sym, 3, rank=r, world=4) for r in range(4)])]],
        # re-hosted build against the hashed set (default), on the fallback lookup, as numbered, the arena path of the mining kernel, shares
        "clique4": [CliqueSolver(dag, 4), CliqueSolver(dag, 4, tune=T(0x800000)), CliqueSolver(dag, 4, tune=T(0x200)), CliqueSolver(dag, 4, tune=T(0x40000)),
                    sum(CliqueSolver(dag, 4, rank=r, world=3) for r in range(3))],
    }
    if use_oracle and g.col_idx.size < 60000000 and kind != 2:  # (dense graphs: the oracle's 4-clique takes minutes)
        osym = O.OGraph(g.row_ptr, g.col_idx); odag = O.orient(osym)
        res["tc"].append(O.tc(odag)); res["diamond"].append(O.diamond(osym)); res["motif3"].append(O.motif3(osym)); res["clique4"].append(O.clique(odag, 4))
    dag.free(); sym.free()
    ok = all(all(x == v[0] for x in v) for v in res.values())
    bad += 0 if ok else 1
    print(f"{case:3d} {name:28s} entries {g.col_idx.size:10d} maxdeg {int(np.diff(g.row_ptr).max()):7d} dag maxdeg {dag.get_max_degree():5d} {'ok' if ok else 'MISMATCH ' + str(res)}", flush=True)
print("cases", n_cases, "mismatches", bad, "seconds", round(time.time() - t0, 1))
sys.exit(1 if bad else 0)
