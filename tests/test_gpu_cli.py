"""The C++ command-line surface (tc_/sgl_/clique_|kcl_/motif_ binaries) on the GPU: argv order and the
final result lines must be byte-identical to the reference mains."""
import os
import subprocess
import sys

import pytest

from common import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "graphminer_amd", "bin")


def run(exe, *args, dev=None):
    """dev: developer options of the library, handed over as `--dev NAME=VALUE` (the apps read nothing from the environment)"""
    p = os.path.join(BIN, exe)
    assert os.path.exists(p), f"{p} not built (make -C graphminer_amd)"
    extra = [a for k, v in (dev or {}).items() for a in ("--dev", f"{k}={v}")]
    r = subprocess.run([p, *map(str, args), *extra], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout.strip().splitlines()


@pytest.mark.parametrize("name", ["citeseer", "cora"])
def test_cli_result_lines(name):
    e = GOLDEN[name]
    prefix = os.path.join(ROOT, "tests", "fixtures", name, "graph")
    out = run("tc_gpu_base", prefix)
    assert out[0] == "Triangle Counting: we assume the neighbor lists are sorted."
    assert f"|V|: {e['nv']}, |E|: {e['dag_ne']}, Max Degree: {e['dag_max_degree']}" in out  # post-orientation meta
    assert out[-1] == f"total_num_triangles = {e['tc']}"
    out = run("sgl_gpu_base", prefix, "diamond")
    assert "Pattern: diamond" in out and out[-1] == f"total_num = {e['diamond']}"
    for pat in ("rectangle", "house", "pentagon", "tailedtriangle", "4path", "3star"):
        assert run("sgl_gpu_base", prefix, pat)[-1] == f"total_num = {e[pat]}"
    out = run("sgl_gpu_base", prefix, "foo")
    assert out[-2:] == ["Not implemented", "total_num = 0"]  # src/sgl/omp_base.cc:51-53
    out = run("clique_gpu_base", prefix, 4)
    assert out[-1] == f"num_4-cliques = {e['clique4']}"
    out = run("kcl_gpu_base", prefix, 4)
    assert f"total_num_cliques = {e['clique4']}" in out  # Pangolin spelling
    out = run("motif_gpu_base", prefix, 3)
    assert out[-2:] == [f"pattern 0: {e['motif3'][0]}", f"pattern 1: {e['motif3'][1]}"]
    assert "num_patterns: 2" in out
    out = run("motif_gpu_base", prefix, 4)
    assert out[-6:] == [f"pattern {i}: {c}" for i, c in enumerate(e["motif4"])] and "num_patterns: 6" in out


def test_cli_unsorted_neighbor_lists(tmp_path):
    """tc_* <graph> 1 1024 0: adj_sorted = 0 -> Graph::sort_neighbors (src/triangle/main.cc:22), here a segmented sort on the GPU"""
    import numpy as np

    from common import load_graph

    g = load_graph("citeseer")
    rng = np.random.default_rng(1)
    col = g.col_idx.copy()
    for v in range(g.V()):
        a, b = int(g.row_ptr[v]), int(g.row_ptr[v + 1])
        col[a:b] = rng.permutation(col[a:b])
    from graphminer_amd import Graph

    Graph(row_ptr=g.row_ptr, col_idx=col).save(str(tmp_path / "graph"))
    out = run("tc_gpu_base", str(tmp_path / "graph"), 1, 1024, 0)
    assert "Sorting the neighbor lists (used for pattern mining)" in out
    assert out[-1] == f"total_num_triangles = {GOLDEN['citeseer']['tc']}"


def test_clique_k5_on_a_row_beyond_4096_and_loud_failures(tmp_path):
    """round 2 refused k >= 5 on DAG rows beyond 4096 entries (and round 1 printed `num_5-cliques = 0` with exit code 0);
    now such a row is counted (cliquek_count_sub_any) -- the reference's kernels have no width limit. What still fails (k out of
    range) prints the reference's "Not implemented yet" and no result line (src/clique/cpu_kernels/automine_omp.h:179-182: exit(0))."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from graphminer_amd.rmat import csr_from_pairs

    W = 4200
    rng = np.random.default_rng(3)
    a, b = np.triu_indices(W, 1)
    keep = rng.random(a.size) < 0.03
    s = np.concatenate([np.zeros(W, dtype=np.uint64), np.repeat(np.arange(1, W + 1, dtype=np.uint64), W - 1), (a[keep] + 1).astype(np.uint64)])
    d = np.concatenate([np.arange(1, W + 1, dtype=np.uint64), np.arange(W + 1, W + 1 + W * (W - 1), dtype=np.uint64), (b[keep] + 1).astype(np.uint64)])
    g = csr_from_pairs(int(W + 1 + W * (W - 1)), s, d)
    g.save(str(tmp_path / "graph"))
    want = O.clique(O.orient(O.OGraph(g.row_ptr, g.col_idx)), 5)
    assert want > 0
    r = subprocess.run([os.path.join(BIN, "clique_gpu_base"), str(tmp_path / "graph"), "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    assert r.stdout.strip().splitlines()[-1] == f"num_5-cliques = {want}"
    r = subprocess.run([os.path.join(BIN, "clique_gpu_base"), str(tmp_path / "graph"), "13"], capture_output=True, text=True, timeout=300)
    assert "Not implemented yet" in r.stdout and "num_13-cliques" not in r.stdout


def test_cli_usage_exits_1():
    r = subprocess.run([os.path.join(BIN, "tc_gpu_base")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage:")


def test_rccl_path_on_one_gpu():
    """broadcast + ncclAllReduce(uint64, sum) code path, forced with a single device"""
    e = GOLDEN["citeseer"]
    prefix = os.path.join(ROOT, "tests", "fixtures", "citeseer", "graph")
    dev = {"GM_FORCE_RCCL_PATH": "1"}
    assert run("tc_multigpu", prefix, 1, dev=dev)[-1] == f"total_num_triangles = {e['tc']}"
    assert run("clique_multigpu", prefix, 4, 1, dev=dev)[-1] == f"num_4-cliques = {e['clique4']}"
    out = run("motif_multigpu", prefix, 3, 1, dev=dev)
    assert out[-2:] == [f"pattern 0: {e['motif3'][0]}", f"pattern 1: {e['motif3'][1]}"]
    assert run("sgl_multigpu", prefix, "diamond", 1, dev=dev)[-1] == f"total_num = {e['diamond']}"
    for pat in ("rectangle", "house", "pentagon"):  # (two kernels each add into the device counter the all-reduce takes)
        assert run("sgl_multigpu", prefix, pat, 1, dev=dev)[-1] == f"total_num = {e[pat]}"
    for pat in ("tailedtriangle", "4path", "3star"):  # four raw per-edge sums all-reduced, then the closed form (gm_sgl4_partial / _finish)
        assert run("sgl_multigpu", prefix, pat, 1, dev=dev)[-1] == f"total_num = {e[pat]}"
    # the environment is NOT a switch: the same variable there changes nothing (one device -> the one-GPU path, no broadcast line)
    r = subprocess.run([os.path.join(BIN, "tc_multigpu"), prefix, "1"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GM_FORCE_RCCL_PATH="1"))
    assert r.returncode == 0 and "RCCL broadcast" not in r.stdout and r.stdout.strip().splitlines()[-1] == f"total_num_triangles = {e['tc']}"
    # asking for more GPUs than present clamps (the reference would fail in cudaSetDevice)
    assert run("tc_multigpu", prefix, 8)[-1] == f"total_num_triangles = {e['tc']}"
