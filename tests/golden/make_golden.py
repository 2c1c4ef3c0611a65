#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference binaries (oracle/_ref/*).

Run in the build container (where /root/reference exists and oracle/ref/Makefile has built
oracle/_ref/):   python tests/golden/make_golden.py

For every graph it runs tc_omp_base, sgl_omp_base {diamond,rectangle,house,pentagon},
clique_omp_base k=4,5 (+ clique_omp_recursive k=6,7 for the k>=6 goldens), motif_omp_base k=3,4
and stores the printed counts. Graphs: the two data fixtures (tests/fixtures/{citeseer,cora}) and
seeded R-MAT graphs produced by graphminer_amd.rmat (only their parameters + a SHA-256 of the
CSR arrays are stored; the graphs are regenerated deterministically by the tests).
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graphminer_amd.graph import Graph  # noqa: E402
from graphminer_amd.rmat import rmat_csr_numpy  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
RMATS = [  # (scale, edge_factor, seed, heavy patterns too? -- True, False, or "sgl": only house / pentagon of the heavy ones)
    (6, 4, 1, True),
    (8, 8, 42, True),
    (10, 16, 42, True),
    (12, 8, 7, "sgl"),
    (14, 16, 42, "sgl"),
]


def run(exe, *args):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    return subprocess.run([os.path.join(REF, exe), *map(str, args)], check=True, capture_output=True, text=True,
                          env=env).stdout


def last_int(out, pat):
    m = re.findall(pat, out)
    assert m, out
    return int(m[-1])


def counts_for(prefix, heavy=True):
    r = {}
    r["tc"] = last_int(run("tc_omp_base", prefix), r"total_num_triangles = (\d+)")
    pats = ["diamond", "rectangle"] + (["house", "pentagon"] if heavy else [])  # (heavy == "sgl" is truthy)
    heavy = heavy is True
    for p in pats:
        r[p] = last_int(run("sgl_omp_base", prefix, p), r"total_num = (\d+)")
    for k in (4, 5):
        r[f"clique{k}"] = last_int(run("clique_omp_base", prefix, k), rf"num_{k}-cliques = (\d+)")
    r["kcl4"] = last_int(run("kcl_omp_base", prefix, 4), r"total_num_cliques = (\d+)")
    if heavy:
        for k in (6, 7):
            r[f"clique{k}"] = last_int(run("clique_omp_recursive", prefix, k), rf"num_{k}-cliques = (\d+)")
    out = run("motif_omp_base", prefix, 3)
    r["motif3"] = [int(x) for x in re.findall(r"pattern \d+: (\d+)", out)]
    if heavy:
        out = run("motif_omp_base", prefix, 4)
        r["motif4"] = [int(x) for x in re.findall(r"pattern \d+: (\d+)", out)]
    # DAG meta as printed by the reference after orientation (graph.cc:646)
    out = run("tc_omp_base", prefix)
    m = re.search(r"\|V\|: (\d+), \|E\|: (\d+), Max Degree: (\d+)", out)
    r["dag_ne"], r["dag_max_degree"] = int(m.group(2)), int(m.group(3))
    return r


def csr_sha(g: Graph):
    h = hashlib.sha256()
    h.update(g.row_ptr.astype("<i8").tobytes())
    h.update(g.col_idx.astype("<i4").tobytes())
    return h.hexdigest()


def main():
    gold = {}
    for name in ("citeseer", "cora"):
        prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
        g = Graph(prefix)
        gold[name] = {"kind": "fixture", "nv": g.V(), "ne": g.E(), "max_degree": g.max_degree, "csr_sha256": csr_sha(g),
                      **counts_for(prefix)}
        print(name, gold[name])
    with tempfile.TemporaryDirectory() as td:
        for scale, ef, seed, heavy in RMATS:
            g = rmat_csr_numpy(scale, ef, seed)
            d = os.path.join(td, g.name)
            os.makedirs(d)
            prefix = os.path.join(d, "graph")
            g.save(prefix)
            gold[g.name] = {"kind": "rmat", "scale": scale, "edge_factor": ef, "seed": seed, "nv": g.V(), "ne": g.E(),
                            "max_degree": g.max_degree, "csr_sha256": csr_sha(g), **counts_for(prefix, heavy)}
            print(g.name, gold[g.name])
    # README known answers for the full-size graphs (no data here; kept for when real files are supplied)
    gold["_readme_known_answers"] = {
        "livej": {"tc": 285730264, "diamond": 76354588342, "motif3": [6412312961, 285730264], "clique4": 9933532019},
        "com-orkut": {"tc": 627584181, "diamond": 67098889426, "motif3": [43742714028, 627584181], "clique4": 3221946137,
                      "clique5": 15766607860},
    }
    with open(os.path.join(ROOT, "tests", "golden", "golden.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
