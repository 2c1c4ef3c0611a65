#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the REAL reference binaries (oracle/_ref/*).

Run in the build container (where /root/reference exists and oracle/ref/Makefile has built
oracle/_ref/):   python tests/golden/make_golden.py

For every graph it runs tc_omp_base, sgl_omp_base {diamond,rectangle,house,pentagon},
clique_omp_base k=4,5 (+ clique_omp_recursive k=6,7,8 for the k>=6 goldens), motif_omp_base k=3,4
and stores the printed counts. EXTRA lists graphs that only get selected (affordable) entries, e.g. the 4-motif
vector of R-MAT-16 from motif_omp_formula.

    python tests/golden/make_golden.py            regenerate everything (hours: house / pentagon of R-MAT-14)
    python tests/golden/make_golden.py --update   keep what golden.json already holds, run only what is missing Graphs: the two data fixtures (tests/fixtures/{citeseer,cora}) and
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


class Lazy(dict):
    """dict that only evaluates (runs the reference binary) for keys golden.json does not hold yet (--update)"""

    def __init__(self, have):
        super().__init__(have or {})

    def put(self, key, fn):
        if key not in self:
            self[key] = fn()
            print("   ran", key, "->", self[key], flush=True)


def extra_counts_for(prefix, what, have=None):
    """selected entries only (EXTRA graphs): `what` = list of keys"""
    r = Lazy(have)
    for key in what:
        if key == "motif4":  # motif_omp_formula: the reference's formula-based 4-motif (src/motif/omp_formula.cc)
            r.put(key, lambda: [int(x) for x in re.findall(r"pattern \d+: (\d+)", run("motif_omp_formula", prefix, 4))])
        elif key == "motif3":
            r.put(key, lambda: [int(x) for x in re.findall(r"pattern \d+: (\d+)", run("motif_omp_base", prefix, 3))])
        elif key == "tc":
            r.put(key, lambda: last_int(run("tc_omp_base", prefix), r"total_num_triangles = (\d+)"))
        elif key == "diamond":
            r.put(key, lambda: last_int(run("sgl_omp_base", prefix, "diamond"), r"total_num = (\d+)"))
        elif key.startswith("clique"):
            k = int(key[6:])
            exe = "clique_omp_base" if k <= 5 else "clique_omp_recursive"
            r.put(key, lambda: last_int(run(exe, prefix, k), rf"num_{k}-cliques = (\d+)"))
        else:
            raise KeyError(key)
    return dict(r)


def counts_for(prefix, heavy=True, have=None):
    if have is not None:
        return counts_for_update(prefix, heavy, have)
    r = {}
    r["tc"] = last_int(run("tc_omp_base", prefix), r"total_num_triangles = (\d+)")
    pats = ["diamond", "rectangle", "tailedtriangle", "4path", "3star"] + (["house", "pentagon"] if heavy else [])  # (heavy == "sgl" is truthy)
    heavy = heavy is True
    for p in pats:
        r[p] = last_int(run("sgl_omp_base", prefix, p), r"total_num = (\d+)")
    for k in (4, 5):
        r[f"clique{k}"] = last_int(run("clique_omp_base", prefix, k), rf"num_{k}-cliques = (\d+)")
    r["kcl4"] = last_int(run("kcl_omp_base", prefix, 4), r"total_num_cliques = (\d+)")
    if True:  # (heavy or not: the recursive solver is quick at every k)
        for k in (6, 7, 8):
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


def counts_for_update(prefix, heavy, have):
    """--update: only the entries golden.json lacks (round 2: clique6..8; round 3: the 4-motif vector of the `sgl` graphs
    R-MAT-12 / -14 from motif_omp_formula, the reference's formula solver -- seconds, where motif_omp_base takes hours)"""
    r = Lazy(have)
    for k in (6, 7, 8):
        r.put(f"clique{k}", lambda: last_int(run("clique_omp_recursive", prefix, k), rf"num_{k}-cliques = (\d+)"))
    r.put("motif4", lambda: [int(x) for x in re.findall(r"pattern \d+: (\d+)", run("motif_omp_formula", prefix, 4))])
    # round 6: k = 9..12 from the reference's generic solver (clique_omp_recursive; its automine / GPU solvers stop at 8), on the graphs
    # where it takes seconds
    if Graph(prefix).E() < 100000:
        for k in (9, 10, 11, 12):
            r.put(f"clique{k}", lambda k=k: last_int(run("clique_omp_recursive", prefix, k), rf"num_{k}-cliques = (\d+)"))
    # round 6: the other 4-vertex SgL patterns of src/sgl/omp_base.cc:21-31 (tailedtriangle.h, 4path.h, 3star.h)
    for pat in ("tailedtriangle", "4path", "3star"):
        r.put(pat, lambda pat=pat: last_int(run("sgl_omp_base", prefix, pat), r"total_num = (\d+)"))
    return dict(r)


def csr_sha(g: Graph):
    h = hashlib.sha256()
    h.update(g.row_ptr.astype("<i8").tobytes())
    h.update(g.col_idx.astype("<i4").tobytes())
    return h.hexdigest()


README_KNOWN = {
    "mico": {"tc": 12534960, "rectangle": 2016507139, "diamond": 3527170461, "house": 1655449692098, "pentagon": 394942854039,
             "clique4": 514864225, "clique5": 19246558419, "clique6": 631568259280, "motif3": [53546459, 12534960],
             "motif4": [2307847995, 4070868075, 3591944265, 33929353, 437985111, 514864225]},
    "patent_citations": {"tc": 6913764, "rectangle": 293116828, "diamond": 75851456, "house": 6586768851, "pentagon": 3254769712,
                         "clique4": 3310556, "clique5": 2976152, "clique6": 3132860, "clique7": 1870484, "clique8": 515317,
                         "motif3": [267600153, 6913764], "motif4": [5148841859, 5764763466, 497680804, 227197040, 55988120, 3310556]},
    "cit-Patents": {"tc": 7515023, "rectangle": 341906226, "diamond": 83785566, "house": 7375094981, "pentagon": 3663584163, "clique4": 3501071},
    "youtube": {"tc": 103017122, "rectangle": 1642566152, "diamond": 1806302028, "house": 71503929498, "pentagon": 24702570492,
                "clique4": 176614367, "clique5": 295551667, "motif3": [1867293654, 103017122],
                "motif4": [201577267737, 55176204040, 8209274276, 366107225, 746615826, 176614367]},
    "livej": {"tc": 285730264, "rectangle": 51520572777, "diamond": 76354588342, "house": 53552979463652, "pentagon": 13892452066046,
              "motif3": [6412312961, 285730264], "clique4": 9933532019, "clique5": 467429836174, "clique6": 20703476954640,
              "motif4": [6619009156172, 1147811961320, 124769176079, 4966580492, 16753396228, 9933532019]},
    "com-orkut": {"tc": 627584181, "rectangle": 127533170575, "diamond": 67098889426, "motif3": [43742714028, 627584181],
                  "clique4": 3221946137, "clique5": 15766607860, "clique6": 75249427585, "clique7": 353962921685, "clique8": 1632691821296,
                  "motif4": [97824018291804, 18573723211463, 1510018661295, 70100119560, 47767212604, 3221946137]},
    "twitter20": {"tc": 17295646010, "diamond": 41166070788458, "clique4": 2123679707619, "clique5": 262607691785539,
                  "motif3": [1780251390046, 17295646010]},
    "twitter40": {"tc": 34824916864, "diamond": 176266103582254, "clique4": 6622234180319, "motif3": [123331114814249, 34824916864]},
    "friendster": {"tc": 4173724142, "rectangle": 465803364346, "diamond": 185191258870, "clique4": 8963503263, "clique5": 21710817218,
                   "clique6": 59926510355, "clique7": 296858496789, "clique8": 3120447373827, "motif3": [708133792538, 4173724142],
                   "motif4": [247358335700296, 364700730542912, 5787076338289, 307502615265, 131410239292, 8963503263]},
    "uk2007": {"tc": 286701284103, "motif3": [25162884716555, 286701284103]},
}

EXTRA = [  # (scale, edge_factor, seed, entries): graphs too big for the full set
    (16, 16, 42, ["tc", "motif3", "motif4"]),
]


def main():
    update = "--update" in sys.argv
    path = os.path.join(ROOT, "tests", "golden", "golden.json")
    old = json.load(open(path)) if update else {}
    gold = {}
    for name in ("citeseer", "cora"):
        prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
        g = Graph(prefix)
        have = old.get(name) if update else None
        gold[name] = {"kind": "fixture", "nv": g.V(), "ne": g.E(), "max_degree": g.max_degree, "csr_sha256": csr_sha(g),
                      **counts_for(prefix, True, have)}
        print(name, gold[name])
    with tempfile.TemporaryDirectory() as td:
        for scale, ef, seed, heavy in RMATS:
            g = rmat_csr_numpy(scale, ef, seed)
            d = os.path.join(td, g.name)
            os.makedirs(d)
            prefix = os.path.join(d, "graph")
            g.save(prefix)
            have = old.get(g.name) if update else None
            gold[g.name] = {"kind": "rmat", "scale": scale, "edge_factor": ef, "seed": seed, "nv": g.V(), "ne": g.E(),
                            "max_degree": g.max_degree, "csr_sha256": csr_sha(g), **counts_for(prefix, heavy, have)}
            print(g.name, gold[g.name])
        for scale, ef, seed, what in EXTRA:
            g = rmat_csr_numpy(scale, ef, seed)
            d = os.path.join(td, g.name)
            os.makedirs(d)
            prefix = os.path.join(d, "graph")
            g.save(prefix)
            have = old.get(g.name) if update else None
            gold[g.name] = {"kind": "rmat", "partial": True, "scale": scale, "edge_factor": ef, "seed": seed, "nv": g.V(),
                            "ne": g.E(), "max_degree": g.max_degree, "csr_sha256": csr_sha(g),
                            **extra_counts_for(prefix, what, have)}
            print(g.name, gold[g.name])
    # README known answers of the public datasets (no data in the image; checked when real files are supplied through
    # GM_DATA_DIR/<name>/graph.*, e.g. converted with graphminer_amd/bin/edgelist2bin). Sources: src/triangle/README.md:52-62,
    # src/sgl/README.md:52-62, src/clique/README.md:54-62, src/motif/README.md:52-60. motif4 in the solver's order
    # [3-star, 4-path, tailed-triangle, 4-cycle, diamond, 4-clique]; rectangle = edge-induced 4-cycles of sgl.
    gold["_readme_known_answers"] = README_KNOWN
    with open(path, "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
