"""Triangle count: the triangles of the HUB CORNER on the matrix cores (gm_ctc.hip) -- the out-edges of the last H vertices of the
topologically renumbered DAG as one masked bit-matrix product (FP4 MFMA), every other edge through the key stream.  The total must equal
the oracle's (omp_base.cc:15-21) for every corner size: none, a few rows, a size that is not a multiple of 64 / 256 (the guarded walk),
the aligned fast path, the whole graph.

GM_TC_CORE_H / GM_TOPO_MIN_ROW are read when a handle's renumbered copy and key stream are built: every case uploads a fresh graph."""
import numpy as np
import pytest

import oracle as O
from graphminer_amd import MotifSolver, TCSolver
from graphminer_amd.rmat import csr_from_pairs, rmat_csr_numpy
from graphminer_amd.solvers import tc_core_info

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available()
    return 0


def _run(g, dev, want, h_expected=None, worlds=(2, 3)):
    with g.to_device(dev) as s, s.orient() as dag:
        got, st = TCSolver(dag, return_stats=True)
        info = tc_core_info(dag)
        assert got == want, (got, want, info)
        assert st.tasks == dag.E()
        if h_expected is not None:
            assert info["h"] == h_expected, info
        assert TCSolver(dag) == want  # again: the dequeue word and the counters are zeroed by every launch
        assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x20000000]) == want  # the kernels without the stream: every edge a task of the lists
        for world in worlds:  # a rank takes every world-th block of the product and its share of the chunks
            parts = [TCSolver(dag, rank=r, world=world, return_stats=True) for r in range(world)]
            assert sum(c for c, _ in parts) == want, (world, [c for c, _ in parts])
            assert sum(t.tasks for _, t in parts) == dag.E()
        return info


@pytest.mark.parametrize("h", [0, 64, 100, 256, 1000, 1024, 4096, 1 << 14])
def test_every_corner_size_counts_the_oracle_triangles(dev, h, devopt):
    devopt("GM_TOPO_MIN_ROW", "0")  # renumber whatever the mean row
    devopt("GM_TC_CORE_H", str(h))
    g = rmat_csr_numpy(14, 24, seed=11 + h)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.tc(O.orient(osym))
    info = _run(g, dev, want)
    nv = g.V()
    if h == 0:
        assert info["h"] == 0
    else:
        assert 0 < info["h"] <= min(h, nv) and info["edges"] > 0 and info["blocks"] >= 1
        assert (info["core_h"] - info["h"]) % 32 == 0  # the corner starts at a word of the bitmap's rows


@pytest.mark.parametrize("nv_odd", [777, 2050, 5001])
def test_corner_of_a_graph_whose_size_is_no_multiple_of_anything(dev, nv_odd, devopt):
    """core bitmap rows of an odd number of words, the whole graph inside the corner (H >= nv)"""
    devopt("GM_TOPO_MIN_ROW", "0")
    devopt("GM_TC_CORE_H", "32768")
    rng = np.random.default_rng(nv_odd)
    m = nv_odd * 40
    s = rng.integers(0, nv_odd, m).astype(np.uint64)
    d = (rng.integers(0, nv_odd, m) ** 2 // nv_odd).astype(np.uint64)  # skewed targets: hubs
    g = csr_from_pairs(nv_odd, s, d)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.tc(O.orient(osym))
    info = _run(g, dev, want)
    assert info["h"] > 0


def test_complete_graph_inside_the_corner(dev, devopt):
    """K_n: every block of the product is full -- C(n, 3)"""
    devopt("GM_TOPO_MIN_ROW", "0")
    devopt("GM_TC_CORE_H", "512")
    n = 512
    iu, ju = np.triu_indices(n, 1)
    g = csr_from_pairs(n, iu.astype(np.uint64), ju.astype(np.uint64))
    info = _run(g, dev, n * (n - 1) * (n - 2) // 6, h_expected=512, worlds=(2,))
    assert info["edges"] == n * (n - 1) // 2


def test_default_rule_takes_a_corner_on_rmat20_and_the_formula_motif_follows(dev):
    """no switches: R-MAT-20 has long rows (renumbered) and a dense hub corner -> the default rule picks one (the largest power-of-two part
    of the core bitmap of >= 3 % density); 3-motif's formula solver counts its triangles with the same launch"""
    g = rmat_csr_numpy(20, 8, seed=42)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.tc(O.orient(osym))
    with g.to_device(dev) as s, s.orient() as dag:
        assert TCSolver(dag) == want
        info = tc_core_info(dag)
        assert info["h"] in (1024, 2048, 4096, 8192, 16384, 32768) and info["core_h"] == 32768, info
        assert info["edges"] >= 0.03 * info["h"] ** 2 / 2
        wedges_tri = MotifSolver(s, 3, formula=True)
        assert wedges_tri[1] == want
        assert MotifSolver(s, 3) == wedges_tri


def _dense_random_graph(n, p, seed):
    rng = np.random.default_rng(seed)
    s, d = np.triu_indices(n, 1)
    keep = rng.random(s.size) < p
    return csr_from_pairs(n, s[keep].astype(np.uint64), d[keep].astype(np.uint64))


@pytest.mark.parametrize("n,p,h", [(2000, 0.9, 1024), (2000, 0.9, 2000), (2300, 0.95, 1024)])
def test_corner_beside_the_two_stage_tables_and_rows_beyond_the_stage(dev, n, p, h, devopt):
    """dense graphs whose DAG rows reach 1025 .. 2048 entries (two task tables: the 1024- and the 2048-entry kernel, forced on this small
    graph) and, for n = 2300, rows beyond the 2048-entry stage -- there the corner must stay OFF (those rows' out-edges are the chunked
    kernel's / sup_long_kernel's): triangle count against the oracle, diamond against the per-edge kernels (one intersection of the symmetric
    lists per edge: no corner, no triangle pass; the oracle's diamond takes minutes on these graphs) with a corner forced"""
    from graphminer_amd import SglSolver

    devopt("GM_TOPO_MIN_ROW", "0")
    devopt("GM_TCT_SPLIT_ALWAYS", "1")
    devopt("GM_TC_CORE_H", str(h))
    devopt("GM_SUP_CORE_H", "1024")
    g = _dense_random_graph(n, p, n + h)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    dmax = int(np.diff(odag.row_ptr).max())
    assert (dmax > 2048) == (n == 2300) and dmax > 1024
    want_tc = O.tc(odag)
    with g.to_device(dev) as s, s.orient() as dag:
        assert TCSolver(dag) == want_tc
        info = tc_core_info(dag)
        assert (info["h"] == 0) == (n == 2300), info
        assert sum(TCSolver(dag, rank=r, world=3) for r in range(3)) == want_tc
        assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x20000000]) == want_tc
        want_dia = SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x10000000])
        assert want_dia > 0 and SglSolver(s, "diamond") == want_dia
        assert SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x40000000]) == want_dia
