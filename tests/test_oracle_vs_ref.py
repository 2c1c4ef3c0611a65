"""Differential test: CPU oracle vs the REAL reference binaries (oracle/_ref, built from
/root/reference by oracle/ref/Makefile) on freshly seeded graphs. Skipped where the binaries
are absent (they travel to the GPU box as prebuilt files, so normally present)."""
import os
import re

import numpy as np
import pytest

import oracle as O
from common import random_graph
from graphminer_amd.rmat import rmat_csr_numpy

pytestmark = pytest.mark.skipif(O.ref_binary("tc_omp_base") is None, reason="oracle/_ref not built")


def _last_int(lines, pat):
    for ln in reversed(lines):
        m = re.search(pat, ln)
        if m:
            return int(m.group(1))
    raise AssertionError(lines)


@pytest.mark.parametrize("case", [("rmat", 9, 8, 3), ("rmat", 11, 6, 11), ("rand", 500, 4000, 5), ("rand", 3000, 9000, 6)])
def test_all_solvers_agree_with_reference(tmp_path, case):
    if case[0] == "rmat":
        g = rmat_csr_numpy(case[1], case[2], case[3])
    else:
        g = random_graph(case[1], case[2], case[3])
    if not (0 < g.max_degree < g.V()):
        pytest.skip("reference loader asserts 0 < max_degree < nv")
    prefix = str(tmp_path / "g" / "graph")
    g.save(prefix)
    sym = O.OGraph(g.row_ptr, g.col_idx)
    dag = O.orient(sym)
    assert O.tc(dag) == _last_int(O.run_ref("tc_omp_base", prefix), r"total_num_triangles = (\d+)")
    assert O.diamond(sym) == _last_int(O.run_ref("sgl_omp_base", prefix, "diamond"), r"total_num = (\d+)")
    assert O.rectangle(sym) == _last_int(O.run_ref("sgl_omp_base", prefix, "rectangle"), r"total_num = (\d+)")
    for k in (4, 5):
        assert O.clique(dag, k) == _last_int(O.run_ref("clique_omp_base", prefix, k), rf"num_{k}-cliques = (\d+)")
    out = O.run_ref("motif_omp_base", prefix, 3)
    assert O.motif3(sym) == [int(re.search(r": (\d+)", ln).group(1)) for ln in out if ln.startswith("pattern")]


def test_oracle_cli_prints_the_reference_result_line(tmp_path):
    """config 1: tc_omp_base on citeseer -> 'total_num_triangles = 1166' (CPU-only plumbing)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "bin", "tc_omp_base")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "bin/tc_omp_base"], stdout=subprocess.DEVNULL)
    prefix = os.path.join(root, "tests", "fixtures", "citeseer", "graph")
    mine = subprocess.run([exe, prefix], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    ref = O.run_ref("tc_omp_base", prefix)
    assert mine[-1] == ref[-1] == "total_num_triangles = 1166"
    assert mine[0] == ref[0]  # banner line, src/triangle/main.cc:13
