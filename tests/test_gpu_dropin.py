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


def run(exe, *args, dev_options=""):
    """dev_options: "NAME=VALUE,..." for the *_hip_seamtest binaries (the same link + integration/dev_options_env.cc, which hands
    GM_DEV_OPTIONS to gm_dev_option before main; the drop-in binaries themselves read nothing from the environment)"""
    p = os.path.join(REF, exe)
    if not os.path.exists(p):
        pytest.skip(f"{exe} not built (needs /root/reference at build time)")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    if dev_options:
        env["GM_DEV_OPTIONS"] = dev_options
    r = subprocess.run([p, *map(str, args)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("name", ["citeseer", "cora"])
def test_reference_mains_with_hip_solvers(name):
    e = GOLDEN[name]
    prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
    hip, omp = run("tc_hip_base", prefix), run("tc_omp_base", prefix)
    assert hip[-1] == omp[-1] == f"total_num_triangles = {e['tc']}"
    for pat in ("diamond", "rectangle", "house", "pentagon", "tailedtriangle", "4path", "3star"):
        hip, omp = run("sgl_hip_base", prefix, pat), run("sgl_omp_base", prefix, pat)
        assert hip[-1] == omp[-1] == f"total_num = {e[pat]}"
    for k in (4, 5):
        hip, omp = run("clique_hip_base", prefix, k), run("clique_omp_base", prefix, k)
        assert hip[-1] == omp[-1] == f"num_{k}-cliques = {e[f'clique{k}']}"
    for k, n in ((3, 2), (4, 6)):
        hip, omp = run("motif_hip_base", prefix, k), run("motif_omp_base", prefix, k)
        assert hip[-n:] == omp[-n:]


@pytest.mark.parametrize("name", ["citeseer", "cora"])
def test_reference_mains_with_hip_solvers_multigpu_seam(name):
    """The n_gpu argument of the reference's mains reaches the n-GPU runner (graphminer_amd/host/multi.cc) through the reference's own
    Graph class: `clique_hip_multigpu <graph> 4 <n_gpu> <chunk>` = the seam of the reference's clique_multigpu (src/clique/multigpu.cu:20),
    `tc_hip_multigpu <graph> <n_gpu> <chunk>` = tc_multigpu_base's. One GPU here: the developer option GM_FORCE_RCCL_PATH (through the *_hip_seamtest links) drives the RCCL path (communicator,
    ncclBroadcast of the CSR, ncclAllReduce of the count) with one rank; a request for more GPUs than present is clamped with a message."""
    e = GOLDEN[name]
    prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
    force = "GM_FORCE_RCCL_PATH=1"
    for ngpu in (1, 2):
        hip = run("clique_hip_seamtest", prefix, 4, ngpu, 256, dev_options=force)
        assert hip[-1] == run("clique_omp_base", prefix, 4)[-1] == f"num_4-cliques = {e['clique4']}"
        assert any("RCCL broadcast" in l for l in hip) and any(l.startswith("runtime[gpu0]") for l in hip), hip
        hip = run("tc_hip_seamtest", prefix, ngpu, 128, dev_options=force)
        assert hip[-1] == f"total_num_triangles = {e['tc']}" and any("RCCL broadcast" in l for l in hip)
    hip = run("sgl_hip_seamtest", prefix, "diamond", 1, dev_options=force)  # diamond across "ranks": supports + reduce-scatter + all-reduce with one rank
    assert hip[-1] == f"total_num = {e['diamond']}" and any("RCCL broadcast" in l for l in hip)
    hip = run("motif_hip_seamtest", prefix, 4, 1, dev_options=force)
    assert hip[-6:] == run("motif_omp_base", prefix, 4)[-6:]
    # the drop-in binaries proper ignore the environment: same variable, one device -> the one-GPU path
    hip = run("tc_hip_multigpu", prefix, 1, 128, dev_options=force)
    assert hip[-1] == f"total_num_triangles = {e['tc']}" and not any("RCCL broadcast" in l for l in hip)
    # without the switch and with one device the same binaries take the one-GPU path
    hip = run("clique_hip_multigpu", prefix, 4)
    assert hip[-1] == f"num_4-cliques = {e['clique4']}" and not any("RCCL broadcast" in l for l in hip)
