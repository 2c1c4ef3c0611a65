"""The drop-in boundary, end to end: the REFERENCE's own main.cc + graph.cc + VertexSet.cc (loader, orientation,
CLI) linked with this repo's solver objects (integration/hip_solvers.cc -> libgraphminer_amd.so) instead of
omp_base.o. Built by oracle/ref/Makefile into oracle/_ref/*_hip_base (where /root/reference exists; the binaries
travel to the GPU box). Their final lines must equal those of the reference's *_omp_base binaries byte for byte."""
import os
import subprocess

import pytest

from common import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
REF = os.path.join(ROOT, "oracle", "_ref")


def run(exe, *args):
    p = os.path.join(REF, exe)
    if not os.path.exists(p):
        pytest.skip(f"{exe} not built (needs /root/reference at build time)")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run([p, *map(str, args)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("name", ["citeseer", "cora"])
def test_reference_mains_with_hip_solvers(name):
    e = GOLDEN[name]
    prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
    hip, omp = run("tc_hip_base", prefix), run("tc_omp_base", prefix)
    assert hip[-1] == omp[-1] == f"total_num_triangles = {e['tc']}"
    for pat in ("diamond", "rectangle", "house", "pentagon"):
        hip, omp = run("sgl_hip_base", prefix, pat), run("sgl_omp_base", prefix, pat)
        assert hip[-1] == omp[-1] == f"total_num = {e[pat]}"
    for k in (4, 5):
        hip, omp = run("clique_hip_base", prefix, k), run("clique_omp_base", prefix, k)
        assert hip[-1] == omp[-1] == f"num_{k}-cliques = {e[f'clique{k}']}"
    for k, n in ((3, 2), (4, 6)):
        hip, omp = run("motif_hip_base", prefix, k), run("motif_omp_base", prefix, k)
        assert hip[-n:] == omp[-n:]
