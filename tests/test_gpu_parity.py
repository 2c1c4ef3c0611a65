"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the golden vectors of
the real reference, and size-independent properties. Run on the MI355X box: pytest -m gpu."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle as O
from common import GOLDEN, GRAPH_NAMES, MotifSolverE, load_graph, random_graph
from graphminer_amd import CliqueSolver, DeviceGraph, Graph, MotifSolver, SglSolver, TCSolver, _lib
from graphminer_amd._lib import dev_option
from graphminer_amd.rmat import csr_from_pairs, rmat_csr_numpy

pytestmark = pytest.mark.gpu


def _complete_graph(n):
    s, d = np.triu_indices(n, 1)
    return csr_from_pairs(n, s.astype(np.uint64), d.astype(np.uint64))


def _star_plus(n_leaves, extra_seed=3):
    """hub 0 joined to everything (row >> staging capacity) + a sparse random graph among the leaves"""
    rng = np.random.default_rng(extra_seed)
    s = np.concatenate([np.zeros(n_leaves, dtype=np.uint64), rng.integers(1, n_leaves + 1, 4 * n_leaves).astype(np.uint64)])
    d = np.concatenate([np.arange(1, n_leaves + 1, dtype=np.uint64), rng.integers(1, n_leaves + 1, 4 * n_leaves).astype(np.uint64)])
    return csr_from_pairs(n_leaves + 1, s, d)


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    lib = _lib.load()
    nfail = C.c_int(0)
    rc = lib.gm_selftest(0, C.byref(nfail))
    assert rc == 0 and nfail.value == 0, f"wave primitive self test failed: {nfail.value} lanes"
    return 0


@pytest.fixture(scope="module", params=GRAPH_NAMES)
def gg(request, dev):
    name = request.param
    g = load_graph(name)
    sym = g.to_device(dev)
    dag = sym.orient()
    yield name, g, sym, dag
    sym.free()
    dag.free()


def test_selftest(dev):
    assert dev == 0


def test_orientation_bit_exact(gg):
    name, g, sym, dag = gg
    want = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    got = dag.download()
    assert np.array_equal(got.row_ptr, want.row_ptr)
    assert np.array_equal(got.col_idx, want.col_idx)
    assert (dag.E(), dag.get_max_degree()) == (GOLDEN[name]["dag_ne"], GOLDEN[name]["dag_max_degree"])


def test_tc_matches_reference(gg):
    name, _, _, dag = gg
    total, st = TCSolver(dag, return_stats=True)
    assert total == GOLDEN[name]["tc"]
    assert st.tasks == dag.E()  # "edges processed" = |E+| (src/triangle/gpu_base.cu:69)
    assert CliqueSolver(dag, 3) == GOLDEN[name]["tc"]
    # (default: the shorter list of every edge streamed against the chunk's rows as one hashed (row, id) set, gm_tch.hip; 0x4000000: the
    # chunked mining kernel that streams N+(v) of every out-edge -- an independent kernel; 0x800000: the hashed set's global-memory fallback)
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x4000000]) == GOLDEN[name]["tc"]
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x800000]) == GOLDEN[name]["tc"]


@pytest.mark.parametrize("tune", [
    [64, 1, 0, 0, 0, 0], [256, 4, 0, 0, 0, 1], [1024, 2, 0, 0, 0, 0], [128, 8, 1, 1, 0, 0], [256, 4, 1, 30, 2, 0],
    [0, 0, 0, 0, 0, 0, 0x1000], [0, 0, 0, 0, 0, 0, 0x4000], [0, 0, 0, 0, 0, 0, 0x3000],
    [256, 4, 8, 1, 0, 0],
    [64, 1, 0, 0, 0, 0, 0x4000000], [256, 4, 1, 30, 2, 0, 0x4000000], [0, 0, 0, 0, 0, 0, 0x4000000 | 0x1000], [128, 8, 1, 1, 0, 0, 0x4000000 | 0x4000],
    [256, 4, 8, 1, 0, 0, 0x4000000], [0, 0, 0, 0, 0, 0, 0x800000 | 0x1000],
])
def test_tc_invariant_under_tuning(gg, tune):
    name, _, sym, dag = gg
    assert TCSolver(dag, tune=tune) == GOLDEN[name]["tc"]
    assert MotifSolverE(sym, 3, tune=tune) == GOLDEN[name]["motif3"]


def test_heavy_chunks_cut_into_parts(gg):
    """tune[6] & 0x1000 cuts every chunk above 4096 estimated entries into parts (several workgroups share one chunk's
    64-edge batches): counts and the per-rank task totals must not change"""
    name, _, sym, dag = gg
    parts = [0, 0, 0, 0, 0, 0, 0x1000]
    assert SglSolver(sym, "diamond", tune=parts) == GOLDEN[name]["diamond"]
    assert MotifSolver(sym, 4, tune=parts) == GOLDEN[name].get("motif4", MotifSolver(sym, 4))
    for policy in (0, 1, 2):
        res = [TCSolver(dag, rank=r, world=3, policy=policy, tune=parts, return_stats=True) for r in range(3)]
        assert sum(c for c, _ in res) == GOLDEN[name]["tc"]
        assert sum(st.tasks for _, st in res) == dag.E()


def test_diamond_listing_form_matches_reference(gg):
    """the nested / listing form (diamond_nested.cuh: materialised S + count_smaller) against the goldens and the count form"""
    name, _, sym, _ = gg
    if GOLDEN[name]["ne"] > 100000:
        pytest.skip("the wave-per-edge listing form is the parity version: small graphs only")
    assert SglSolver(sym, "diamond", tune=[0, 0, 0, 0, 0, 0, 1024]) == GOLDEN[name]["diamond"]
    assert sum(SglSolver(sym, "diamond", rank=r, world=3, tune=[0, 0, 0, 0, 0, 0, 1024]) for r in range(3)) == GOLDEN[name]["diamond"]


def test_diamond_matches_reference(gg):
    name, _, sym, _ = gg
    total, st = SglSolver(sym, "diamond", return_stats=True)
    assert total == GOLDEN[name]["diamond"]
    assert st.tasks == sym.E() // 2
    assert SglSolver(sym, "diamond", tune=[128, 2, 0, 0, 0, 1]) == GOLDEN[name]["diamond"]
    # one GPU, default: edge supports from the triangles of the DAG (gm_sup.hip), then sum C(t, 2); 0x10000000: one intersection of the
    # two symmetric lists per edge (what several ranks run); 0x800000: the supports' hashed set on its global-memory fallback lookup;
    # 0x200: the DAG as numbered; 0x1000: heavy chunks cut into parts
    P = 0x10000000
    total, st = SglSolver(sym, "diamond", tune=[0, 0, 0, 0, 0, 0, P], return_stats=True)
    assert total == GOLDEN[name]["diamond"] and st.tasks == sym.E() // 2
    for t6 in (0x800000, 0x200, 0x1000, 0x200 | 0x1000, 0x4000, P | 0x1000, P | 0x4000):
        assert SglSolver(sym, "diamond", tune=[0, 0, 0, 0, 0, 0, t6]) == GOLDEN[name]["diamond"], hex(t6)
    assert SglSolver(sym, "diamond", tune=[64, 1, 0, 0, 0, 0]) == GOLDEN[name]["diamond"]


@pytest.mark.parametrize("pattern", ["tailedtriangle", "4path", "3star"])
def test_sgl_four_vertex_patterns_from_the_per_edge_sums(gg, pattern):
    """tailedtriangle.h / 4path.h / 3star.h (src/sgl/omp_base.cc:21-31): the HIP path takes them from the four per-edge sums of the formula
    4-motif (one |N(u) ^ N(v)| per edge, csrc/gm_launch.hip gm_sgl); goldens from sgl_omp_base, the oracle restates the loop nests"""
    name, g, sym, _ = gg
    e = GOLDEN[name]
    if pattern not in e:
        pytest.skip("no golden")
    total, st = SglSolver(sym, pattern, return_stats=True)
    assert total == e[pattern] and st.kernel_ms > 0
    assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x80000]) == e[pattern]  # every row through the general kernel
    assert SglSolver(sym, pattern) == e[pattern]
    if e["ne"] < 100000:
        osym = O.OGraph(g.row_ptr, g.col_idx)
        assert total == {"tailedtriangle": O.tailedtriangle, "4path": O.path4, "3star": O.star3}[pattern](osym)
    # a rank's partial sums cannot be halved / divided by six on their own: refused, like the reference has no multi-GPU SgL
    lib = _lib.load()
    la = _lib.gm_launch()
    la.world, la.rank = 2, 0
    tot = C.c_uint64(7)
    assert lib.gm_sgl(sym.handle, pattern.encode(), C.byref(la), C.byref(tot), None) == _lib.GM_ERR_UNSUPPORTED
    # ... so several ranks add up their raw sums (gm_sgl4_partial) and finish once
    from graphminer_amd.solvers import sgl4_finish, sgl4_partial

    for world in (2, 5):
        raw = [0, 0, 0, 0]
        for r in range(world):
            raw = [x + y for x, y in zip(raw, sgl4_partial(sym, rank=r, world=world))]
        assert sgl4_finish(pattern, raw) == e[pattern]
    assert lib.gm_sgl4_finish(b"diamond", (C.c_uint64 * 4)(), C.byref(tot)) == _lib.GM_ERR_INVALID


@pytest.mark.parametrize("pattern", ["rectangle", "house", "pentagon"])
def test_sgl_nested_patterns_match_reference(gg, pattern):
    """rectangle.h / house.h / pentagon.h loop nests on the wave64 primitives; goldens from sgl_omp_base"""
    name, _, sym, _ = gg
    e = GOLDEN[name]
    if pattern not in e:
        pytest.skip("no golden (pattern too slow for the reference binary at this size)")
    total, st = SglSolver(sym, pattern, return_stats=True)
    assert total == e[pattern]
    # (pentagon's rank partials are two's-complement halves of even sums: exact modulo 2^64, like the 3-motif wedge partials)
    assert sum(SglSolver(sym, pattern, rank=r, world=3, chunk=32) for r in range(3)) % 2**64 == e[pattern]
    assert sum(SglSolver(sym, pattern, rank=r, world=2, policy=1) for r in range(2)) % 2**64 == e[pattern]
    if pattern == "pentagon":  # the other implementations: wedges + per-round flat intersections (on the descending copy / as numbered)
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800]) == e[pattern]
        if e["ne"] < 100000:  # (the wedge form as numbered needs many seconds on R-MAT-14)
            assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800 | 0x200]) == e[pattern]
        assert SglSolver(sym, pattern) == e[pattern]
    assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 512]) == e[pattern]  # on the graph as numbered (no degree renumbering)
    if pattern == "rectangle":  # the other two implementations: wedges + flattened intersections; one wave per edge
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 2048]) == e[pattern]
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 1024]) == e[pattern]
        assert SglSolver(sym, pattern) == e[pattern]  # (the counter maps are left zeroed by every launch)
    if pattern == "house":  # the other implementations: flattened (v0,v1,v3) form with / without the LDS S-bitmap; one wave per edge
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800]) == e[pattern]
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x8000]) == e[pattern]
        assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800 | 0x200]) == e[pattern]
        if e["ne"] < 100000:  # (the wave-per-edge loop nest needs minutes on R-MAT-14)
            assert SglSolver(sym, pattern, tune=[0, 0, 0, 0, 0, 0, 1024]) == e[pattern]
        assert SglSolver(sym, pattern) == e[pattern]  # (the maps are left zeroed by every launch)


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_rectangle_counter_maps_in_lds(dev, name, devopt):
    """rectangle.h:1-11 by wedge accumulation with the heavy centres' counter maps in LDS (gm_mine.hip rect_lds_kernel: (centre, id range)
    tasks, row bounds per range boundary): GM_RECT_LDS_MIN=1 sends EVERY centre there on the golden graphs; goldens from sgl_omp_base"""
    e = GOLDEN[name]
    if "rectangle" not in e:
        pytest.skip("no golden")
    g = load_graph(name)
    devopt("GM_RECT_LDS_MIN", "1")
    with g.to_device(dev) as sym:
        total, st = SglSolver(sym, "rectangle", return_stats=True)
        assert total == e["rectangle"] and st.kernel_ms > 0
        assert SglSolver(sym, "rectangle") == e["rectangle"]  # (the maps are left zeroed by every task)
        assert sum(SglSolver(sym, "rectangle", rank=r, world=3) for r in range(3)) == e["rectangle"]
        assert SglSolver(sym, "rectangle", tune=[0, 0, 0, 0, 0, 0, 0x20000]) == e["rectangle"]  # every end in the global maps (round 5)
        if "motif4" in e:
            assert MotifSolver(sym, 4) == e["motif4"]
    devopt("GM_RECT_LDS_MIN", None)
    with g.to_device(dev) as sym:  # the default threshold
        assert SglSolver(sym, "rectangle") == e["rectangle"]


def test_rectangle_lds_maps_beside_the_global_ones(dev, devopt):
    """a graph of more vertices than the LDS ranges cover (GM_RECT_LDS_RANGES=2: the hubs' 32 K ids and one range of packed counters below
    them): the ends below the cut stay in the global maps, the ranges above it hold every centre's own id somewhere (rows cut at v0) -- the
    two forms of the kernel and the flattened wedge form agree"""
    g = rmat_csr_numpy(19, 4, 3)
    devopt("GM_RECT_LDS_MIN", "1")
    devopt("GM_RECT_LDS_RANGES", "2")
    with g.to_device(dev) as sym:
        a = SglSolver(sym, "rectangle")
        b = SglSolver(sym, "rectangle", tune=[0, 0, 0, 0, 0, 0, 0x20000])
        c = SglSolver(sym, "rectangle", tune=[0, 0, 0, 0, 0, 0, 2048])
        assert a == b == c and a > 0
        assert sum(SglSolver(sym, "rectangle", rank=r, world=4, policy=1) for r in range(4)) == a
    devopt("GM_RECT_LDS_RANGES", None)
    with g.to_device(dev) as sym:  # every id in a range: 32-bit counters at the hubs, packed ones below
        assert SglSolver(sym, "rectangle") == a
    devopt("GM_RECT_LDS_MIN", None)
    with g.to_device(dev) as sym:
        assert SglSolver(sym, "rectangle") == a


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_house_maps_in_lds(dev, name, devopt):
    """house.h:1-16 by wedge accumulation with the heavy centres' (count | weighted sum) maps in LDS (gm_mine.hip house_lds_kernel);
    GM_RECT_LDS_MIN=1 sends every centre there; goldens from sgl_omp_base"""
    e = GOLDEN[name]
    if "house" not in e:
        pytest.skip("no golden")
    g = load_graph(name)
    devopt("GM_RECT_LDS_MIN", "1")
    with g.to_device(dev) as sym:
        total, st = SglSolver(sym, "house", return_stats=True)
        assert total == e["house"] and st.kernel_ms > 0
        assert SglSolver(sym, "house") == e["house"]  # (the maps are left zeroed)
        assert sum(SglSolver(sym, "house", rank=r, world=3) for r in range(3)) % 2**64 == e["house"]
        assert SglSolver(sym, "house", tune=[0, 0, 0, 0, 0, 0, 0x20000]) == e["house"]  # every end in the global maps (round 5)
    devopt("GM_RECT_LDS_RANGES", "1")  # one range of 16 K ids in LDS, the ends below it in the global maps
    with g.to_device(dev) as sym:
        assert SglSolver(sym, "house") == e["house"]
    devopt("GM_RECT_LDS_RANGES", None)
    devopt("GM_RECT_LDS_MIN", None)
    with g.to_device(dev) as sym:  # the default threshold
        assert SglSolver(sym, "house") == e["house"]


def test_house_hub_row_longer_than_lds_bitmap():
    """a hub with 17,000 neighbours (> the 16,384-bit LDS S-bitmap) plus a sparse random graph: the flattened kernel's
    long-row path against the wave-per-edge loop nest (house.h order) on the same graph"""
    rng = np.random.default_rng(5)
    n = 17001
    hub = n - 1  # highest id: every hub edge is a symmetry-broken (v0 = hub, v1) task
    s = np.concatenate([np.full(n - 1, hub), rng.integers(0, n - 1, 40000)]).astype(np.uint64)
    d = np.concatenate([np.arange(n - 1), rng.integers(0, n - 1, 40000)]).astype(np.uint64)
    g = csr_from_pairs(n, s, d)
    assert int(np.diff(g.row_ptr).max()) > 16384
    want = O.house(O.OGraph(g.row_ptr, g.col_idx))  # the restated loop nest of house.h:1-16 (seconds: one hub row)
    with DeviceGraph.upload(g) as sym:
        flat = SglSolver(sym, "house")
        assert flat == want
        assert flat == SglSolver(sym, "house", tune=[0, 0, 0, 0, 0, 0, 1024])
        assert flat == SglSolver(sym, "house", tune=[0, 0, 0, 0, 0, 0, 0x800])
        assert flat == SglSolver(sym, "house", tune=[0, 0, 0, 0, 0, 0, 0x8000])
        # (a rank's partial is exact modulo 2^64: a centre's positive term -- its maps, in LDS -- and its negative ones may be tasks of different ranks)
        assert flat == sum(SglSolver(sym, "house", rank=r, world=3) for r in range(3)) % 2**64


def test_clique4_matches_reference(gg):
    name, _, _, dag = gg
    assert CliqueSolver(dag, 4) == GOLDEN[name]["clique4"]
    assert CliqueSolver(dag, 4, tune=[64, 1, 1, 1, 0, 0]) == GOLDEN[name]["clique4"]
    assert CliqueSolver(dag, 4, tune=[512, 4, 0, 0, 0, 1]) == GOLDEN[name]["clique4"]
    assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x800000]) == GOLDEN[name]["clique4"]  # the build's hashed set on its fallback lookup


@pytest.mark.parametrize("k", [5, 6, 7, 8, 9, 10, 11, 12])
def test_clique_k_matches_reference(gg, k):
    """k = 5 (automine_5clique, automine_omp.h:138-157) and k = 6 .. 12 (goldens from clique_omp_recursive; the reference's GPU solver stops
    at 8, src/clique/gpu_base.cu:59-71) on the same bit-matrix: C_1(S)=|S|, C_m(S)=sum_{j in S} C_{m-1}(S & M_j)."""
    name, _, _, dag = gg
    e = GOLDEN[name]
    if f"clique{k}" not in e:
        pytest.skip("no golden for this k")
    assert CliqueSolver(dag, k) == e[f"clique{k}"]
    if e[f"clique{k}"] > 10**11 or k >= 11:
        return  # (R-MAT-14, k = 8: 133 G cliques, 25 s per run; k = 11, 12 on R-MAT-10: 9 s per run -- the variants below are covered by the smaller graphs / k)
    assert CliqueSolver(dag, k, tune=[64, 1, 0, 0, 0, 1]) == e[f"clique{k}"]
    assert sum(CliqueSolver(dag, k, rank=r, world=3) for r in range(3)) == e[f"clique{k}"]


def test_algorithmic_bytes_clique4(gg):
    """SURVEY 8(d) 4-clique bytes: TC formula (level 1) + gm_clique4_level2_bytes == the oracle's one-pass value"""
    import ctypes as C

    name, g, _, dag = gg
    odag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    l2 = C.c_uint64(0)
    assert _lib.load().gm_clique4_level2_bytes(dag.handle, C.byref(l2)) == 0
    assert O.alg_bytes("tc", odag) + int(l2.value) == O.alg_bytes("clique4", odag)


def test_motif3_matches_reference(gg):
    name, _, sym, _ = gg
    assert MotifSolverE(sym, 3) == GOLDEN[name]["motif3"]  # [wedges, triangles]: CPU order
    # gm_motif's default for k = 3 is the formula solver (the triangles of the DAG, wedges derived); it reports the graph's entries as tasks
    got, st = MotifSolver(sym, 3, return_stats=True)
    assert got == GOLDEN[name]["motif3"] and st.tasks == sym.E()
    parts = [MotifSolver(sym, 3, rank=r, world=4) for r in range(4)]
    assert [sum(p[0] for p in parts) % 2**64, sum(p[1] for p in parts)] == GOLDEN[name]["motif3"]
    # motif_omp_formula variant (src/motif/omp_formula.cc:39-40): identical counts, also when partitioned
    assert MotifSolver(sym, 3, formula=True) == GOLDEN[name]["motif3"]
    parts = [MotifSolver(sym, 3, formula=True, rank=r, world=3) for r in range(3)]
    assert [sum(p[0] for p in parts) % 2**64, sum(p[1] for p in parts)] == GOLDEN[name]["motif3"]


def test_motif4_matches_reference(gg):
    """formula-based 4-motif (automine_formula.h:21-56 + omp_formula.cc:41-45) = [3-star, 4-path, tailed-triangle,
    4-cycle, diamond, 4-clique]; goldens from motif_omp_base (== motif_omp_formula)"""
    import ctypes as C

    name, _, sym, _ = gg
    e = GOLDEN[name]
    if "motif4" not in e:
        pytest.skip("no golden")
    assert MotifSolver(sym, 4) == e["motif4"]
    # multi-rank: raw partial sums add up, then one finish
    lib = _lib.load()
    tot = [0] * 6
    for r in range(3):
        la = _lib.gm_launch()
        la.rank, la.world = r, 3
        raw = (C.c_uint64 * 6)()
        assert lib.gm_motif4_partial(sym.handle, C.byref(la), raw, None) == 0
        tot = [a + int(b) for a, b in zip(tot, raw)]
    out = (C.c_uint64 * 6)()
    assert lib.gm_motif4_finish((C.c_uint64 * 6)(*tot), out) == 0
    assert [int(x) for x in out] == e["motif4"]


@pytest.mark.parametrize("world,policy", [(2, 0), (3, 0), (8, 0), (2, 1), (5, 1), (2, 2), (7, 2)])
def test_task_partition_sums_to_the_whole(gg, world, policy):
    name, _, sym, dag = gg
    e = GOLDEN[name]
    tc = dia = k4 = 0
    m3 = [0, 0]
    tasks = 0
    for r in range(world):
        t, st = TCSolver(dag, rank=r, world=world, policy=policy, return_stats=True)
        tc += t
        tasks += st.tasks
        dia += SglSolver(sym, "diamond", rank=r, world=world, policy=policy)
        k4 += CliqueSolver(dag, 4, rank=r, world=world, policy=policy)
        m = MotifSolverE(sym, 3, rank=r, world=world, policy=policy)
        m3 = [(m3[0] + m[0]) % 2**64, m3[1] + m[1]]  # per-rank wedge partials are modulo 2^64 (like the uint64 all-reduce)
    assert (tc, dia, k4, m3) == (e["tc"], e["diamond"], e["clique4"], e["motif3"])
    assert tasks == dag.E()


def test_unsupported_and_invalid_arguments(gg):
    _, _, sym, dag = gg
    assert SglSolver(sym, "foo") == 0  # "Not implemented", total_num = 0 (src/sgl/omp_base.cc:51-53)
    with pytest.raises(_lib.GraphMinerError):
        CliqueSolver(dag, 2)
    with pytest.raises(_lib.GraphMinerError):
        TCSolver(dag, rank=3, world=2)


# ---- edge cases ----------------------------------------------------------------------------------
def test_empty_and_tiny_graphs(dev):
    g = Graph(row_ptr=[0, 0, 0, 0], col_idx=[]).to_device(dev)
    d = g.orient()
    assert TCSolver(d) == 0 and SglSolver(g, "diamond") == 0 and CliqueSolver(d, 4) == 0 and MotifSolverE(g, 3) == [0, 0]
    # a single edge, a path, a triangle
    for rp, ci, want in [([0, 1, 2], [1, 0], (0, 0, 0, [0, 0])),
                         ([0, 1, 3, 4], [1, 0, 2, 1], (0, 0, 0, [1, 0])),
                         ([0, 2, 4, 6], [1, 2, 0, 2, 0, 1], (1, 0, 0, [0, 1]))]:
        s = Graph(row_ptr=rp, col_idx=ci).to_device(dev)
        o = s.orient()
        assert (TCSolver(o), SglSolver(s, "diamond"), CliqueSolver(o, 4), MotifSolverE(s, 3)) == want


@pytest.mark.parametrize("n", [5, 64, 65, 130])
def test_complete_graph_closed_forms(dev, n):
    g = _complete_graph(n)
    s = g.to_device(dev)
    d = s.orient()
    assert TCSolver(d) == math.comb(n, 3)
    assert CliqueSolver(d, 4) == math.comb(n, 4)
    for k in ((5, 6, 7, 8) if n <= 65 else (5, 6)):
        assert CliqueSolver(d, k) == math.comb(n, k)
    if n == 5:
        assert [CliqueSolver(d, k) for k in (9, 12)] == [0, 0]
        k20 = _complete_graph(20).to_device(dev).orient()
        assert [CliqueSolver(k20, k) for k in (9, 10, 11, 12)] == [math.comb(20, k) for k in (9, 10, 11, 12)]
        assert CliqueSolver(_complete_graph(28).to_device(dev).orient(), 12) == math.comb(28, 12)
    assert SglSolver(s, "diamond") == math.comb(n, 2) * math.comb(n - 2, 2)
    assert MotifSolverE(s, 3) == [0, math.comb(n, 3)]


def test_rows_longer_than_the_lds_staging_capacity(dev):
    """K_1100: every symmetric row (1099) exceeds the 1024-entry stage -> split chunks, HBM search;
    DAG rows up to 1099 -> clique bit-matrix in the global scratch arena."""
    n = 1100
    s = _complete_graph(n).to_device(dev)
    d = s.orient()
    assert d.get_max_degree() == n - 1
    assert TCSolver(d) == math.comb(n, 3)
    assert MotifSolverE(s, 3) == [0, math.comb(n, 3)]
    assert SglSolver(s, "diamond") == math.comb(n, 2) * math.comb(n - 2, 2)
    general = [0, 0, 0, 0, 0, 0, 0x80000]  # (rows > 1024 entries go to the hashed-row class by default; this is the general kernel)
    assert MotifSolverE(s, 3, tune=general) == [0, math.comb(n, 3)]
    assert SglSolver(s, "diamond", tune=general) == math.comb(n, 2) * math.comb(n - 2, 2)
    assert CliqueSolver(d, 4) == math.comb(n, 4)


def test_big_rows_deeper_cliques(dev):
    """K_300 / K_600: DAG rows beyond 256 columns -> scratch-resident matrix; k = 5 compacts the induced sub-matrices (LDS up
    to 256 set bits, a second arena slot beyond), k = 6 keeps the per-sub-tree walk for the wide rows"""
    n = 300
    d = _complete_graph(n).to_device(dev).orient()
    assert CliqueSolver(d, 5) == math.comb(n, 5)
    assert CliqueSolver(d, 5, tune=[0, 0, 0, 0, 0, 0, 0x20]) == math.comb(n, 5)  # the per-sub-tree walk everywhere (A/B)
    assert CliqueSolver(d, 5, tune=[0, 0, 0, 0, 0, 0, 0x200000]) == math.comb(n, 5)  # the any-width pair count (A/B)
    d = _complete_graph(270).to_device(dev).orient()
    assert CliqueSolver(d, 6) == math.comb(270, 6)
    d = _complete_graph(600).to_device(dev).orient()
    assert CliqueSolver(d, 5) == math.comb(600, 5)


def test_deeper_cliques_on_a_row_wider_than_2048(dev):
    """a DAG row of 2100 entries (two words per lane in cliquek_count_sub; round 1 refused k >= 5 beyond 2048). Vertex 0 has
    2100 neighbours w_j, every w_j has 2099 private leaves (so deg(w_j) >= deg(0) and 0 -> w_j in the degree-ordered DAG) and
    the w_j carry a random graph; every k-clique through 0 is a (k-1)-clique of that graph -- checked against the oracle"""
    rng = np.random.default_rng(11)
    W = 2100
    s = [np.zeros(W, dtype=np.uint64)]
    d = [np.arange(1, W + 1, dtype=np.uint64)]
    leaf = W + 1
    ws = np.repeat(np.arange(1, W + 1, dtype=np.uint64), W - 1)
    s.append(ws)
    d.append(np.arange(leaf, leaf + ws.size, dtype=np.uint64))
    a, b = np.triu_indices(W, 1)
    keep = rng.random(a.size) < 0.12
    s.append((a[keep] + 1).astype(np.uint64))
    d.append((b[keep] + 1).astype(np.uint64))
    g = csr_from_pairs(int(leaf + ws.size), np.concatenate(s), np.concatenate(d))
    odag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    assert int(np.diff(odag.row_ptr).max()) == W and int(odag.row_ptr[1] - odag.row_ptr[0]) == W
    dg = g.to_device(dev).orient()
    for k in (4, 5, 6):
        want = O.clique(odag, k)
        assert want > 0
        assert CliqueSolver(dg, k) == want
        if k > 4:
            assert CliqueSolver(dg, k, tune=[0, 0, 0, 0, 0, 0, 0x200000]) == want  # any-width pair count instead of the tile walk
    # beyond 4096 entries (round 2 refused k >= 5 there; the reference's kernels have no width limit, clique5_warp_edge.cuh:3-39): the
    # same construction with a row of 4200 entries -- everything from the workgroup's global scratch (cliquek_count_sub_any)
    W2 = 4200
    a2, b2 = np.triu_indices(W2, 1)
    keep2 = rng.random(a2.size) < 0.04
    s2 = np.concatenate([np.zeros(W2, dtype=np.uint64), np.repeat(np.arange(1, W2 + 1, dtype=np.uint64), W2 - 1), (a2[keep2] + 1).astype(np.uint64)])
    d2 = np.concatenate([np.arange(1, W2 + 1, dtype=np.uint64), np.arange(W2 + 1, W2 + 1 + W2 * (W2 - 1), dtype=np.uint64), (b2[keep2] + 1).astype(np.uint64)])
    g2 = csr_from_pairs(int(W2 + 1 + W2 * (W2 - 1)), s2, d2)
    odag2 = O.orient(O.OGraph(g2.row_ptr, g2.col_idx))
    d2g = g2.to_device(dev).orient()
    assert d2g.get_max_degree() == W2 == int(np.diff(odag2.row_ptr).max())
    for k in (4, 5, 6):
        want = O.clique(odag2, k)
        assert want > 0
        assert CliqueSolver(d2g, k) == want, k
    assert sum(CliqueSolver(d2g, 5, rank=r, world=3) for r in range(3)) == O.clique(odag2, 5)


def _hub_over_a_random_graph(W, p, seed, planted=0):
    """vertex 0 with W neighbours w_j, every w_j with W - 1 private leaves (0 -> w_j in the degree-ordered DAG: a row of W entries), a random
    graph of density p on the w_j and a planted clique on the first `planted` of them"""
    rng = np.random.default_rng(seed)
    a, b = np.triu_indices(W, 1)
    keep = (rng.random(a.size) < p) | (b < planted)
    s = np.concatenate([np.zeros(W, dtype=np.uint64), np.repeat(np.arange(1, W + 1, dtype=np.uint64), W - 1), (a[keep] + 1).astype(np.uint64)])
    d = np.concatenate([np.arange(1, W + 1, dtype=np.uint64), np.arange(W + 1, W + 1 + W * (W - 1), dtype=np.uint64), (b[keep] + 1).astype(np.uint64)])
    return csr_from_pairs(int(W + 1 + W * (W - 1)), s, d)


@pytest.mark.parametrize("W,p,planted", [(600, 0.3, 0), (2100, 0.12, 12), (4200, 0.04, 12)])
def test_cliques_of_nine_to_twelve_on_wide_rows(dev, W, p, planted):
    """k = 9..12 (the reference counts them with its generic clique_omp_recursive / edge_warp_iterative.cuh:2-75; gpu_base.cu:59-71 stops at 8)
    where the deeper levels are CALLED, not inlined (gm_chunk.h kCliqueInlineM): a DAG row of 600 entries (sub-matrices in LDS / the second
    arena slot), of 2100 (two words per lane) and of 4200 (everything from the workgroup's global scratch), against the oracle's DFS"""
    g = _hub_over_a_random_graph(W, p, 23, planted)
    odag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    assert int(np.diff(odag.row_ptr).max()) == W
    dg = g.to_device(dev).orient()
    got = {}
    for k in (9, 10, 11, 12):
        want = O.clique(odag, k)
        got[k] = want
        assert CliqueSolver(dg, k) == want, k
    assert got[9] > 0 and got[10] > 0 and (planted == 0 or got[12] > 0)
    assert sum(CliqueSolver(dg, 9, rank=r, world=3) for r in range(3)) == got[9]
    assert CliqueSolver(dg, 9, tune=[0, 0, 0, 0, 0, 0, 0x20]) == got[9]  # the per-sub-tree walk (A/B)


def test_deeper_cliques_sub_matrix_path_rmat14(dev):
    """R-MAT-14 (ef 16) DAG: induced-sub-matrix path against the per-sub-tree walk, k = 5 and 6"""
    g = rmat_csr_numpy(14, 16, 42)
    d = g.to_device(dev).orient()
    odag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    for k in (5, 6):
        want = GOLDEN[g.name][f"clique{k}"]  # clique_omp_base (k = 5) / clique_omp_recursive (k = 6) of the reference
        assert O.clique(odag, k) == want     # ... and the oracle's DFS on the same DAG
        assert CliqueSolver(d, k) == want
        assert CliqueSolver(d, k, tune=[0, 0, 0, 0, 0, 0, 0x20]) == want
        assert sum(CliqueSolver(d, k, rank=r, world=4) for r in range(4)) == want
    assert CliqueSolver(d, 7) == GOLDEN[g.name]["clique7"]


def _dense_random_graph(n, p, seed):
    rng = np.random.default_rng(seed)
    s, d = np.triu_indices(n, 1)
    keep = rng.random(s.size) < p
    return csr_from_pairs(n, s[keep].astype(np.uint64), d[keep].astype(np.uint64))


@pytest.mark.parametrize("n,p", [(700, 0.6), (1500, 0.5), (2200, 0.9), (2600, 0.9)])
def test_clique4_wide_vertices_two_phases(dev, n, p):
    """dense random graphs: DAG out-degrees in every class of the two-phase wide path (gm_mine.h): count class S (d+ < 512),
    L (whole matrix in 112 KB of LDS, d+ <= 896), X (column blocks, d+ <= 2048; n = 2200) and, for n = 2600, rows beyond 2048
    that stay on the mining kernel's arena path -- against the CPU oracle, against the all-in-the-mining-kernel build of the
    same count (tune[6] & 0x40000), across rank shares, and with an arena so small that the share needs several rounds"""
    import os

    g = _dense_random_graph(n, p, n)
    odag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    dmax = int(np.diff(odag.row_ptr).max())
    assert dmax > {700: 256, 1500: 512, 2200: 1024, 2600: 2048}[n]
    d = g.to_device(dev).orient()
    want = O.clique(odag, 4) if n < 2000 else None
    got = CliqueSolver(d, 4)
    if want is not None:
        assert got == want
    assert CliqueSolver(d, 4, tune=[0, 0, 0, 0, 0, 0, 0x40000]) == got
    assert CliqueSolver(d, 4, tune=[0, 0, 0, 0, 0, 0, 0x800000]) == got  # the build's hashed set on its fallback lookup
    assert sum(CliqueSolver(d, 4, rank=r, world=3) for r in range(3)) == got
    assert sum(CliqueSolver(d, 4, rank=r, world=5, policy=1) for r in range(5)) == got
    assert CliqueSolver(d, 4, tune=[0, 0, 1, 1, 0, 0]) == got  # another direction rule: pass Y on the wide rows too
    dev_option("GM_WIDE_ARENA_MB", "4")  # 4 MiB arena: plans are cached per (rank, world, policy), so use a fresh share
    try:
        assert sum(CliqueSolver(d, 4, rank=r, world=2, policy=1) for r in range(2)) == got
    finally:
        dev_option("GM_WIDE_ARENA_MB", None)


@pytest.mark.parametrize("name", ["citeseer", "rmat12_ef8_s7", "rmat14_ef16_s42"])
def test_big_handle_paths_on_small_graphs(dev, name, devopt):
    """A graph of >= 2^31 entries gets a BIG handle (64-bit offsets: upload, orientation, download, the formula 3-motif; every solver
    that walks the graph itself refuses it). GM_BIG_NE=1 forces that handle for a small graph: the 64-bit orientation kernels must
    produce the DAG of the 32-bit ones bit for bit, and the counts the goldens."""
    g = load_graph(name)
    ref = g.to_device(dev)
    ref_dag = ref.orient().download()
    devopt("GM_BIG_NE", "1")
    big = g.to_device(dev)
    devopt("GM_BIG_NE", None)
    back = big.download()
    assert np.array_equal(back.row_ptr, g.row_ptr) and np.array_equal(back.col_idx, g.col_idx)
    dag = big.orient()
    got = dag.download()
    assert np.array_equal(got.row_ptr, ref_dag.row_ptr) and np.array_equal(got.col_idx, ref_dag.col_idx)
    e = GOLDEN[name]
    assert TCSolver(dag) == e["tc"] and CliqueSolver(dag, 4) == e["clique4"]
    assert MotifSolver(big, 3) == e["motif3"] == MotifSolver(big, 3, formula=True)
    assert SglSolver(big, "diamond") == e["diamond"]  # (edge supports of the oriented copy, gm_sup.hip)
    for bad in (lambda: SglSolver(big, "diamond", rank=0, world=2), lambda: MotifSolver(big, 4), lambda: TCSolver(big), lambda: SglSolver(big, "rectangle")):
        with pytest.raises(_lib.GraphMinerError) as ei:
            bad()
        assert ei.value.status == _lib.GM_ERR_TOO_LARGE
    for h in (dag, big, ref):
        h.free()


@pytest.mark.parametrize("name", GRAPH_NAMES)
def test_dag_patterns_on_the_topological_view(dev, name, devopt):
    """TC / k-clique / the formula 3-motif on the topologically renumbered copy of the DAG (get_relabeled mode 2: trimmed in-edge tasks,
    upper-triangular matrices). The library takes that view only where rows are long (sum d+^2 / |E+| >= 64: none of the small golden
    graphs); GM_TOPO_MIN_ROW=0 forces it. The counts are the goldens; tune[6] & 0x200 runs on the graph as numbered."""
    devopt("GM_TOPO_MIN_ROW", "0")
    g = load_graph(name)
    sym = g.to_device(dev)
    dag = sym.orient()
    e = GOLDEN[name]
    assert TCSolver(dag) == e["tc"]
    assert CliqueSolver(dag, 3) == e["tc"]
    assert CliqueSolver(dag, 4) == e["clique4"]
    assert MotifSolver(sym, 3, formula=True) == e["motif3"]
    assert sum(CliqueSolver(dag, 4, rank=r, world=3) for r in range(3)) == e["clique4"]
    assert sum(TCSolver(dag, rank=r, world=4, policy=1) for r in range(4)) == e["tc"]
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x200]) == e["tc"] and CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x200]) == e["clique4"]
    if "motif4" in e:
        assert MotifSolver(sym, 4) == e["motif4"]
    # the renumbered copy itself: a permutation of the same graph, topological, rows ascending
    dag.free()
    sym.free()


def _planted_dag(seed, perm):
    """a sparse random DAG (edges i -> j, i < j) with PLANTED wide rows: vertices 0..12 get out-degrees in every class of the
    4-clique pipeline -- narrow (<= 256), count classes S / L / X, the stage boundary 1024 / 2048, and one row beyond kCbMaxDeg that
    stays on the mining kernel's arena path -- pointing into a sparse background, so that the CPU oracle stays affordable; edges
    between the planted vertices make in-edge (type B) tasks of every size. perm: the same DAG under a random renumbering (no
    longer topological: the build kernel then streams whole lists)."""
    rng = np.random.default_rng(seed)
    n = 7000
    degs = [200, 256, 257, 400, 511, 513, 700, 896, 1000, 1025, 1500, 2047, 2048, 2300]
    np_ = len(degs)
    s, d = np.triu_indices(n - 100, 1)
    keep = rng.random(s.size) < 0.012  # background among the vertices 100..n
    src, dst = [s[keep] + 100], [d[keep] + 100]
    for u, k in enumerate(degs):
        nb = rng.choice(np.arange(100, n), size=k - (np_ - 1 - u), replace=False)
        src.append(np.full(nb.size, u))
        dst.append(nb)
        src.append(np.full(np_ - 1 - u, u))  # u -> every later planted vertex (longer lists: they host these edges)
        dst.append(np.arange(u + 1, np_))
    src, dst = np.concatenate(src).astype(np.int64), np.concatenate(dst).astype(np.int64)
    if perm:
        pi = rng.permutation(n)
        src, dst = pi[src], pi[dst]
    keys = np.unique((src << 32) | dst)
    src, dst = keys >> 32, (keys & 0xFFFFFFFF).astype(np.int32)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=rp[1:])
    return Graph(row_ptr=rp, col_idx=dst, name=f"planted{seed}{'p' if perm else ''}")


@pytest.mark.parametrize("perm", [False, True])
def test_clique4_planted_wide_rows_against_oracle(dev, perm):
    """VERDICT r2 weak #1a: the count classes S / L / X (and every row width around the stage / class boundaries) against the ORACLE,
    on a hand-built DAG (gm_clique takes any DAG); the re-hosted build (gm_cbuild.hip) with its topological trimming (perm = False) and
    without (perm = True); the all-in-the-mining-kernel build of the same count; rank shares under every policy; several arena rounds"""
    import os

    g = _planted_dag(11, perm)
    odag = O.OGraph(g.row_ptr, g.col_idx)
    assert int(np.diff(g.row_ptr).max()) == 2300
    want = O.clique(odag, 4)
    assert want > 1000
    d = g.to_device(dev)
    got, st = CliqueSolver(d, 4, return_stats=True)
    assert got == want
    assert st.tasks == odag.ne
    assert CliqueSolver(d, 4, tune=[0, 0, 0, 0, 0, 0, 0x40000]) == want
    assert TCSolver(d) == CliqueSolver(d, 3) == O.tc(odag)
    for world, policy in ((3, 0), (4, 1), (2, 2)):
        assert sum(CliqueSolver(d, 4, rank=r, world=world, policy=policy) for r in range(world)) == want, (world, policy)
    dev_option("GM_WIDE_ARENA_MB", "1")  # 1 MiB arena: the narrow chunks and the wide rows need several rounds (fresh share: plans are cached)
    try:
        assert sum(CliqueSolver(d, 4, rank=r, world=2, policy=0, chunk=256) for r in range(2)) == want
    finally:
        dev_option("GM_WIDE_ARENA_MB", None)
    d.free()


def test_hub_graph_against_oracle(dev):
    g = _star_plus(5000)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    s = g.to_device(dev)
    d = s.orient()
    assert TCSolver(d) == O.tc(odag)
    assert MotifSolverE(s, 3) == O.motif3(osym)
    assert SglSolver(s, "diamond") == O.diamond(osym)
    general = [0, 0, 0, 0, 0, 0, 0x80000]  # (the hub row through the general kernel instead of the hashed-row class)
    assert MotifSolverE(s, 3, tune=general) == O.motif3(osym)
    assert SglSolver(s, "diamond", tune=general) == O.diamond(osym)
    assert CliqueSolver(d, 4) == O.clique(odag, 4)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_graphs_against_oracle(dev, seed):
    g = random_graph(2000 * seed, 30000 * seed, seed)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    s = g.to_device(dev)
    d = s.orient()
    assert TCSolver(d) == O.tc(odag)
    assert SglSolver(s, "diamond") == O.diamond(osym)
    assert CliqueSolver(d, 4) == O.clique(odag, 4)
    assert CliqueSolver(d, 5) == O.clique(odag, 5)
    assert MotifSolverE(s, 3) == O.motif3(osym)
    if seed == 1:  # the flattened wedge kernels and the 4-motif formula path, against the restated loop nests
        assert SglSolver(s, "rectangle") == O.rectangle(osym)
        assert SglSolver(s, "house") == O.house(osym)
        assert SglSolver(s, "pentagon") == O.pentagon(osym)
        assert MotifSolver(s, 4) == O.motif4(osym)


# ---- full-size, size-independent properties -------------------------------------------------------
def test_large_rmat_properties(dev):
    """R-MAT scale 18 (ef 16): kernels must agree with each other, with the wedge closed form
    sum_v C(d,2) - 3T (src/motif/cpu_kernels/automine_formula.h:2-19), with the oracle's TC, and be
    invariant under partitioning / staging."""
    from graphminer_amd.rmat import rmat_csr_device

    s, rp, ci = rmat_csr_device(18, 16, 42, dev)
    d = s.orient()
    t = TCSolver(d)
    wedges, tri = MotifSolverE(s, 3)
    assert tri == t == CliqueSolver(d, 3)
    deg = (rp[1:] - rp[:-1]).cpu().numpy().astype(object)
    assert wedges == int(sum(x * (x - 1) // 2 for x in deg)) - 3 * t
    assert sum(TCSolver(d, rank=r, world=4) for r in range(4)) == t
    assert TCSolver(d, tune=[256, 4, 0, 0, 0, 1]) == t
    host = s.download()
    assert O.tc(O.orient(O.OGraph(host.row_ptr, host.col_idx))) == t
    dia = SglSolver(s, "diamond")
    assert sum(SglSolver(s, "diamond", rank=r, world=3, policy=1) for r in range(3)) == dia
    k4 = CliqueSolver(d, 4)
    assert sum(CliqueSolver(d, 4, rank=r, world=2) for r in range(2)) == k4


def test_hub_paths_against_oracle_rmat16(dev):
    """R-MAT-16 (ef 16, max degree 9.6 K): rows longer than the 3072-entry stage -> SPLIT chunks with dense bitmaps, the LDS
    pre-filter, the longer-row-hosts rule and cost-cut parts are all exercised; diamond / 3-motif against the CPU oracle on
    the same graph (4-motif against the oracle's recorded answer), and invariance under the scheduling knobs"""
    g = rmat_csr_numpy(16, 16, 42)
    assert g.max_degree > 3072
    osym = O.OGraph(g.row_ptr, g.col_idx)
    s = g.to_device(dev)
    want_d, want_m3 = O.diamond(osym), O.motif3(osym)
    assert SglSolver(s, "diamond") == want_d
    assert MotifSolverE(s, 3) == want_m3
    # (default: the workgroup classes take the rows > 1024 entries; 0x80000: every row through the general kernel -- SPLIT chunks, dense
    # bitmaps, the LDS pre-filter; 0x100000: the classes forced on, which is also the default here)
    G = 0x80000
    for tune in ([0, 0, 0, 0, 0, 0, 0x1000], [0, 0, 0, 0, 0, 0, 0x4000], [0, 0, 0, 0, 0, 0, 0x100], [0, 0, 0, 0, 0, 0, 0x4], [0, 0, 0, 0, 0, 1],
                 [0, 0, 0, 0, 0, 0, G], [0, 0, 0, 0, 0, 0, G | 0x1000], [0, 0, 0, 0, 0, 0, G | 0x4000], [0, 0, 0, 0, 0, 0, G | 0x100], [0, 0, 0, 0, 0, 0, G | 0x4],
                 [0, 0, 0, 0, 0, 0, 0x100000], [0, 0, 0, 0, 0, 0, 0x100000 | 0x1000], [0, 0, 0, 0, 0, 0, 0x100000 | 0x4000],
                 # (0x400000: the classes with the sorted LDS copy + bisection instead of the hashed set; 0x800000: the hashed-set
                 # kernels with every lookup through their global-memory fallback)
                 [0, 0, 0, 0, 0, 0, 0x100000 | 0x400000], [0, 0, 0, 0, 0, 0, 0x100000 | 0x400000 | 0x1000], [0, 0, 0, 0, 0, 0, 0x100000 | 0x800000],
                 [0, 0, 0, 0, 0, 0, 0x100000 | 0x2000000]):  # (0x2000000: the hash with the 32-bit multiply of id spaces > 2^24)
        assert SglSolver(s, "diamond", tune=tune) == want_d
        assert SglSolver(s, "diamond", tune=(tune + [0])[:6] + [(tune + [0] * 7)[6] | 0x10000000]) == want_d  # (the per-edge kernels on one GPU)
        assert MotifSolverE(s, 3, tune=tune) == want_m3
    assert sum(SglSolver(s, "diamond", rank=r, world=8) for r in range(8)) == want_d
    assert sum(SglSolver(s, "diamond", rank=r, world=8, tune=[0, 0, 0, 0, 0, 0, G]) for r in range(8)) == want_d
    parts = [MotifSolverE(s, 3, rank=r, world=3, tune=[0, 0, 0, 0, 0, 0, G]) for r in range(3)]
    assert [sum(p[i] for p in parts) % 2**64 for i in range(2)] == want_m3
    assert MotifSolver(s, 4, tune=[0, 0, 0, 0, 0, 0, G]) == GOLDEN[g.name]["motif4"]
    assert sum(SglSolver(s, "diamond", rank=r, world=5, tune=[0, 0, 0, 0, 0, 0, 0x100000]) for r in range(5)) == want_d
    parts = [MotifSolverE(s, 3, rank=r, world=4, tune=[0, 0, 0, 0, 0, 0, 0x100000]) for r in range(4)]
    assert [sum(p[i] for p in parts) % 2**64 for i in range(2)] == want_m3
    assert MotifSolver(s, 4, tune=[0, 0, 0, 0, 0, 0, 0x100000]) == GOLDEN[g.name]["motif4"]
    assert MotifSolver(s, 4, tune=[0, 0, 0, 0, 0, 0, 0x100000 | 0x400000]) == GOLDEN[g.name]["motif4"]
    parts = [MotifSolverE(s, 3, rank=r, world=3, policy=2) for r in range(3)]
    assert [sum(p[i] for p in parts) % 2**64 for i in range(2)] == want_m3
    # golden.json: tc / motif3 / motif4 of this graph from the reference's tc_omp_base, motif_omp_base, motif_omp_formula
    e = GOLDEN[g.name]
    assert e["csr_sha256"] == __import__("common").csr_sha(g)
    assert want_m3 == e["motif3"] and TCSolver(s.orient()) == e["tc"]
    assert MotifSolver(s, 4) == e["motif4"]


def test_hashed_row_classes_on_adversarial_ids(dev):
    """The hashed-set kernels (gm_hrow.hip) on rows built against their hash: ids = h * C^-1 mod 2^K land in chosen buckets.
    Hub 1 (1500 entries): five buckets take 20 ids each -> 12 surplus ids per bucket in the surplus list; hub 2 (1300 entries in
    21 buckets) and hub 3 (9000 entries, every bucket of its class-2 table twice over) overflow the list -> the global-memory
    lookup; hub 4 (30000 entries) is a giant row of two pieces; hub 5 (5000 random ids) the ordinary case. Against the CPU
    oracle, classes forced on (0x100000), and every variant of the class kernels."""
    K = 16
    nv = 1 << K
    ck = ((0x9E3779B97F4A7C15 >> (64 - K)) | 1) & (nv - 1)
    inv = pow(ck, -1, nv)
    rng = np.random.default_rng(7)
    hubs = [3, 40001, 17, 5, 60000]

    def ids_of_hashes(h):
        x = (np.asarray(h, dtype=np.uint64) * np.uint64(inv)) & np.uint64(nv - 1)
        return x

    # hub 1: n = 1500 -> LB = 9 (512 buckets of 2^7 hash values): 5 buckets x 20 + 1400 spread over the others
    sh1 = K - 9
    h1 = np.concatenate([np.arange(20, dtype=np.uint64) + (np.uint64(b) << np.uint64(sh1)) for b in (7, 100, 101, 300, 511)] +
                        [(np.arange(512, dtype=np.uint64) << np.uint64(sh1)) + np.uint64(64 + j) for j in range(3)])[:1500]
    # hub 2: n = 1300 -> LB = 9: 21 buckets, 62 ids each
    h2 = np.concatenate([np.arange(62, dtype=np.uint64) + (np.uint64(b * 11) << np.uint64(sh1)) for b in range(21)])
    # hub 3: n = 9000 -> class 2, LB = 12 (4096 buckets of 16 hash values): 563 buckets completely full
    h3 = np.arange(9000, dtype=np.uint64) + np.uint64(4096)
    s, d = [], []
    for hub, hs in zip(hubs[:3], (h1, h2, h3)):
        x = ids_of_hashes(hs)
        s.append(np.full(x.size, hub, dtype=np.uint64)); d.append(x)
    x4 = rng.choice(nv, 30000, replace=False).astype(np.uint64)
    s.append(np.full(x4.size, hubs[3], dtype=np.uint64)); d.append(x4)
    x5 = rng.choice(nv, 5000, replace=False).astype(np.uint64)
    s.append(np.full(x5.size, hubs[4], dtype=np.uint64)); d.append(x5)
    # background: partner lists of 5..60 keys, some of them long (vertices 100..131 get ~3000 neighbours each)
    s.append(rng.integers(0, nv, 600000).astype(np.uint64)); d.append(rng.integers(0, nv, 600000).astype(np.uint64))
    for v in range(100, 132):
        y = rng.choice(nv, 3000, replace=False).astype(np.uint64)
        s.append(np.full(y.size, v, dtype=np.uint64)); d.append(y)
    g = csr_from_pairs(nv, np.concatenate(s), np.concatenate(d))
    deg = np.diff(g.row_ptr)
    assert deg[hubs[3]] > 24576 and 8191 < deg[hubs[2]] <= 24576 and 1024 < deg[hubs[0]] <= 8191
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want_d, want_m3 = O.diamond(osym), O.motif3(osym)
    sd = g.to_device(dev)
    general = [0, 0, 0, 0, 0, 0, 0x80000]
    assert SglSolver(sd, "diamond", tune=general) == want_d and MotifSolverE(sd, 3, tune=general) == want_m3
    want_m4 = MotifSolver(sd, 4, tune=general)
    # (0x2000000: the kernels instantiated for id spaces beyond 2^24 -- v_mul_lo_u32 instead of v_mul_u32_u24 in the hash)
    for flags in (0x100000, 0x100000 | 0x800000, 0x100000 | 0x400000, 0x100000 | 0x1000000, 0x100000 | 0x1000, 0x100000 | 0x2000000):
        tune = [0, 0, 0, 0, 0, 0, flags]
        assert SglSolver(sd, "diamond", tune=tune) == want_d, hex(flags)
        assert MotifSolverE(sd, 3, tune=tune) == want_m3, hex(flags)
        assert MotifSolver(sd, 4, tune=tune) == want_m4, hex(flags)
    assert sum(SglSolver(sd, "diamond", rank=r, world=3, tune=[0, 0, 0, 0, 0, 0, 0x100000]) for r in range(3)) == want_d
    parts = [MotifSolverE(sd, 3, rank=r, world=5, tune=[0, 0, 0, 0, 0, 0, 0x100000]) for r in range(5)]
    assert [sum(p[i] for p in parts) % 2**64 for i in range(2)] == want_m3


def test_tc_rows_beyond_the_task_list_stage(dev):
    """A dense core inside a sparse graph: DAG rows beyond the 2048-entry stage of tct_kernel host nothing, their out-edges go through
    the chunked kernel (own table), every other edge through the task lists -- against the CPU oracle, both A/B paths, rank shares."""
    rng = np.random.default_rng(5)
    nv, core = 40000, 2400
    cs, cd = np.triu_indices(core, 1)
    keep = rng.random(cs.size) < 0.93
    s = np.concatenate([cs[keep], rng.integers(0, nv, 300000)]).astype(np.uint64)
    d = np.concatenate([cd[keep], rng.integers(0, nv, 300000)]).astype(np.uint64)
    g = csr_from_pairs(nv, s, d)
    dag = g.to_device(dev).orient()
    assert dag.get_max_degree() > 2048
    want = O.tc(O.orient(O.OGraph(g.row_ptr, g.col_idx)))
    got, st = TCSolver(dag, return_stats=True)
    assert got == want and st.tasks == dag.E()
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x4000000]) == want  # the chunked kernel alone
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x4000000]) == want  # the chunked mining kernel
    assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x800000]) == want   # ... against the hashed set's fallback lookup
    assert sum(TCSolver(dag, rank=r, world=3) for r in range(3)) == want
    assert sum(TCSolver(dag, rank=r, world=4, policy=2) for r in range(4)) == want
    assert CliqueSolver(dag, 3) == want


def test_differential_fuzz_across_implementation_paths(dev):
    """scripts/exp/fuzz_paths.py: 24 random graphs (R-MAT, hub, dense, flat), every pattern through its default path and through the
    alternative implementations behind the A/B switches, plus rank shares -- all counts equal"""
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "scripts", "exp", "fuzz_paths.py"), "24"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches 0" in r.stdout


@pytest.mark.parametrize("pattern", ["rectangle", "house", "pentagon"])
def test_sgl_map_kernels_match_flat_kernels_rmat14(dev, pattern):
    """R-MAT-14 (ef 16, max degree 3.6 K: heavy centres use all four waves on one map): the wedge-accumulation kernels against
    the flattened-intersection kernels (independent implementations), as numbered and on the renumbered copies, and across
    rank shares"""
    g = rmat_csr_numpy(14, 16, 42)
    s = g.to_device(dev)
    want = SglSolver(s, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800])
    if pattern in GOLDEN[g.name]:  # from the reference's sgl_omp_base (house: 25 min on 8 threads)
        assert want == GOLDEN[g.name][pattern]
    assert SglSolver(s, pattern) == want
    assert SglSolver(s, pattern, tune=[0, 0, 0, 0, 0, 0, 0x200]) == want
    assert SglSolver(s, pattern, tune=[0, 0, 0, 0, 0, 0, 0x800 | 0x200]) == want
    assert sum(SglSolver(s, pattern, rank=r, world=8) for r in range(8)) % 2**64 == want
    assert sum(SglSolver(s, pattern, rank=r, world=3, policy=1) for r in range(3)) % 2**64 == want
    assert SglSolver(s, pattern) == want


def test_sort_neighbors_on_the_device(dev):
    """Graph::sort_neighbors (src/common/graph.cc:138-146; tc_* with adj_sorted = 0): rows shuffled on the host, sorted by
    one segmented radix sort on the device, equal to the original CSR; the solvers then see the golden counts"""
    g = load_graph("rmat12_ef8_s7")
    rng = np.random.default_rng(3)
    shuffled = g.col_idx.copy()
    for v in range(g.V()):
        a, b = int(g.row_ptr[v]), int(g.row_ptr[v + 1])
        shuffled[a:b] = rng.permutation(shuffled[a:b])
    assert not np.array_equal(shuffled, g.col_idx)
    u = Graph(row_ptr=g.row_ptr.copy(), col_idx=shuffled)
    with u.to_device(dev) as s:
        s.sort_neighbors()
        back = s.download()
        assert np.array_equal(back.col_idx, g.col_idx) and np.array_equal(back.row_ptr, g.row_ptr)
        d = s.orient()
        assert TCSolver(d) == GOLDEN[g.name]["tc"]
        assert MotifSolverE(s, 3) == GOLDEN[g.name]["motif3"]
        with pytest.raises(_lib.GraphMinerError):  # (after a solver has built its tables the rows must not move any more)
            s.sort_neighbors()


def test_rmat_device_generator_equals_numpy(dev):
    from graphminer_amd.rmat import rmat_csr_device

    for scale, ef, seed in [(10, 16, 42), (12, 8, 7)]:
        s, rp, ci = rmat_csr_device(scale, ef, seed, dev)
        g = rmat_csr_numpy(scale, ef, seed)
        assert np.array_equal(rp.cpu().numpy(), g.row_ptr)
        assert np.array_equal(ci.cpu().numpy(), g.col_idx)
        assert TCSolver(s.orient()) == GOLDEN[g.name]["tc"]


def test_tc_hashed_set_with_colliding_ids(dev):
    """gm_tch.hip keeps a chunk's DAG rows as a hashed set of four-slot buckets (bucket = top bits of id * 0x9E3779B1, XOR a row salt).
    Ids chosen through the inverse of that multiplier all fall into ONE bucket per row. They are made the high-degree side of the graph,
    so that they are what the DAG rows hold: rows of 300 of them overflow the surplus list (the whole chunk then bisects its rows in
    global memory), rows with 6 of them among 40 ordinary ids use the surplus list. Counts against the CPU oracle and the sorted-copy
    kernel, on the graph as numbered (0x200: these ids reach the kernel) and on the topological copy."""
    rng = np.random.default_rng(11)
    nv = 1 << 24
    cinv = pow(0x9E3779B1, -1, 1 << 32)
    t = (np.uint64(5) << np.uint64(22)) + np.arange(1 << 22, dtype=np.uint64)
    x = (t * np.uint64(cinv)) & np.uint64(0xFFFFFFFF)
    one_bucket = np.sort(x[(x < nv) & (x > 100000)])
    assert one_bucket.size > 5000
    plain = rng.choice(np.arange(50000, 100000, dtype=np.uint64), size=400, replace=False)
    for pool_size, nhub, per, extra in ((600, 2000, 300, 0), (600, 3000, 6, 40)):
        pool = one_bucket[:pool_size]
        hubs = np.arange(1, nhub + 1, dtype=np.uint64)
        s = [np.repeat(hubs, per)]
        d = [np.concatenate([rng.choice(pool, size=per, replace=False) for _ in hubs])]
        if extra:
            s.append(np.repeat(hubs, extra))
            d.append(np.concatenate([rng.choice(plain, size=extra, replace=False) for _ in hubs]))
        both = np.concatenate([pool, plain]) if extra else pool
        s.append(rng.choice(both, size=30000))  # edges among the high-degree vertices, so that triangles exist
        d.append(rng.choice(both, size=30000))
        g = csr_from_pairs(int(nv), np.concatenate(s), np.concatenate(d))
        dag = g.to_device(dev).orient()
        got = dag.download()
        rows = np.diff(got.row_ptr)[1:nhub + 1]
        assert rows.min() >= per  # the hubs' DAG rows do hold the colliding ids
        want = O.tc(O.orient(O.OGraph(g.row_ptr, g.col_idx)))
        assert want > 0
        assert TCSolver(dag) == want
        assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x200]) == want
        assert TCSolver(dag, tune=[0, 0, 0, 0, 0, 0, 0x200 | 0x4000000]) == want
        assert sum(TCSolver(dag, rank=r, world=3, tune=[0, 0, 0, 0, 0, 0, 0x200]) for r in range(3)) == want
        # the re-hosted 4-clique build keeps the same rows as a hashed (row, id) -> position set (gm_cbuild.hip)
        want4 = O.clique(O.orient(O.OGraph(g.row_ptr, g.col_idx)), 4)
        assert CliqueSolver(dag, 4) == want4
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x200]) == want4
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x200 | 0x800000]) == want4  # ... on its global-memory fallback lookup
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x40000]) == want4           # the mining kernel's arena path


@pytest.mark.parametrize("row,x", [(1024, 3461295), (1023, 3461295), (2048, 1364143), (2047, 1364143)])
def test_hashed_position_set_rejects_the_empty_slot_indices(dev, row, x):
    """in gm_hset.h an EMPTY slot / the overflow MARKER XOR the probe word decode to the stage indices STAGE - 1 /
    STAGE - 2 for every key whose hash has its low 32 - LB bits all ones (3461295 for the 1024-entry stage, 1364143 for 2048) -- a
    host row that covers those indices took such a key for a member.  Vertex 0 gets a DAG row of exactly `row` hubs, hub 1 also points
    at x (not a neighbour of 0): the task (0 -> 1) streams x against the row.  Diamond (edge supports) and 4-clique (re-hosted build)
    against the oracle and against the kernels that do not use the position set."""
    rng = np.random.default_rng(row)
    hubs = np.arange(1, row + 1, dtype=np.int64)
    iu, ju = np.triu_indices(row, 1)
    keep = rng.random(iu.size) < 0.02
    s, d = [np.zeros(row, dtype=np.int64), hubs[iu[keep]]], [hubs, hubs[ju[keep]]]
    s.append(np.array([1, 2, 3], dtype=np.int64))  # x beside three hubs, so that the task lists that hold it are short
    d.append(np.full(3, x, dtype=np.int64))
    s, d = np.concatenate(s), np.concatenate(d)
    deg = np.bincount(np.concatenate([s, d]), minlength=x + 1)
    want_deg = row + rng.integers(0, 17, size=row)  # hubs out-rank vertex 0 (degree `row`, smallest id), in a mixed order among themselves
    nxt, ls, ld = row + 1, [], []
    for v, t in list(zip(hubs, want_deg)) + [(x, row + 8)]:
        n = int(t - deg[v])
        ids = np.arange(nxt, nxt + n, dtype=np.int64)
        ids = ids + (ids >= x)  # (leaf ids skip x)
        nxt += n
        ls.append(np.full(n, v, dtype=np.int64))
        ld.append(ids)
    s, d = np.concatenate([s] + ls), np.concatenate([d] + ld)
    nv = int(max(x + 1, d.max() + 1))
    g = csr_from_pairs(nv, s.astype(np.uint64), d.astype(np.uint64))
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    assert int(odag.row_ptr[1] - odag.row_ptr[0]) == row and x not in set(g.col_idx[g.row_ptr[0]:g.row_ptr[1]].tolist())
    sym = g.to_device(dev)
    dag = sym.orient()
    per_edge = [0, 0, 0, 0, 0, 0, 0x10000000]
    want_d, want_4 = O.diamond(osym), O.clique(odag, 4)
    assert SglSolver(sym, "diamond", tune=per_edge) == want_d
    assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x40000]) == want_4
    for t6 in (0, 0x200):
        assert SglSolver(sym, "diamond", tune=[0, 0, 0, 0, 0, 0, t6]) == want_d, hex(t6)
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, t6]) == want_4, hex(t6)
    assert TCSolver(dag) == O.tc(odag)
    sym.free()


def test_unsorted_rows_are_refused_until_sorted(dev):
    """every solver relies on ascending rows (bisection, positions, the trimmed tasks of a topologically numbered DAG). A
    hand-built DAG (every edge to a larger id) whose rows are stored DESCENDING used to pass the topological check (first entry of each
    row above its vertex) and lose triangles silently; now the first solver call checks the rows once per handle and refuses, and
    gm_graph_sort_neighbors (the reference's adj_sorted = 0 path) makes the handle usable."""
    g = _planted_dag(5, False)
    odag = O.OGraph(g.row_ptr, g.col_idx)
    want3, want4 = O.tc(odag), O.clique(odag, 4)
    rev = g.col_idx.copy()
    for v in range(g.V()):
        a, b = int(g.row_ptr[v]), int(g.row_ptr[v + 1])
        rev[a:b] = rev[a:b][::-1]
    u = Graph(row_ptr=g.row_ptr.copy(), col_idx=rev, name="planted5_descending")
    with u.to_device(dev) as d:
        for call in (lambda: TCSolver(d), lambda: CliqueSolver(d, 4), lambda: TCSolver(d, tune=[0, 0, 0, 0, 0, 0, 0x200])):
            with pytest.raises(_lib.GraphMinerError) as ei:
                call()
            assert ei.value.status == _lib.GM_ERR_INVALID and b"ascending" in _lib.load().gm_last_error()
        d.sort_neighbors()
        assert TCSolver(d) == want3 and CliqueSolver(d, 4) == want4
        assert TCSolver(d, tune=[0, 0, 0, 0, 0, 0, 0x200]) == want3 and CliqueSolver(d, 4, tune=[0, 0, 0, 0, 0, 0, 0x200]) == want4


@pytest.mark.parametrize("world", [2, 3, 8])
def test_diamond_supports_of_rank_shares_add_up(gg, world, dev):
    """the several-rank diamond (gm_diamond_support_partial / _finish): ONE GPU plays every rank -- each rank's share of the triangle pass
    into its own support array, the arrays summed (what the reduce-scatter does), every rank's slice through sum C(t, 2): the parts add up
    to the reference's count; the sum of the arrays is the one-rank support array."""
    import torch

    from graphminer_amd.solvers import diamond_support_finish, diamond_support_partial, diamond_support_size

    name, g, s, d = gg
    want = GOLDEN[name]["diamond"]
    n = diamond_support_size(s, world)
    assert n % (64 * world) == 0 and n >= d.E()
    bufs = [torch.full((n,), 7, dtype=torch.int32, device=f"cuda:{dev}") for _ in range(world)]  # (garbage in: the call zeroes its buffer)
    for r in range(world):
        diamond_support_partial(s, bufs[r].data_ptr(), n, rank=r, world=world)
    total = torch.stack(bufs).sum(0, dtype=torch.int64).to(torch.int32)
    one = torch.empty(diamond_support_size(s, 1), dtype=torch.int32, device=f"cuda:{dev}")
    diamond_support_partial(s, one.data_ptr(), one.numel())
    assert torch.equal(total[:d.E()], one[:d.E()]) and int(total[d.E():].abs().sum()) == 0
    assert int(total.sum()) == 3 * GOLDEN[name]["tc"]  # three increments per triangle
    per = n // world
    parts = [diamond_support_finish(s, total[r * per:(r + 1) * per].contiguous().data_ptr(), per) for r in range(world)]
    assert sum(parts) == want, (world, parts)
    assert SglSolver(s, "diamond") == want


def test_diamond_supports_with_rows_beyond_the_stage(dev):
    """a DAG row of more than 2048 entries fits no hashed set: the triangle count sends its out-edges to the chunked kernel, and until
    round 4 the edge supports gave up (gm_sgl fell back to one intersection per edge; a handle of >= 2^31 entries got GM_ERR_TOO_LARGE).
    sup_long_kernel takes those edges now -- one wave per edge, bisection in global memory: diamond from the supports against the per-edge
    kernels, the oracle, and rank shares through gm_diamond_support_partial / _finish."""
    import torch

    from graphminer_amd.solvers import diamond_support_finish, diamond_support_partial, diamond_support_size

    g = _dense_random_graph(2300, 0.95, 23)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    assert int(np.diff(odag.row_ptr).max()) > 2048
    want = O.diamond(osym)
    with g.to_device(dev) as s:
        assert SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x10000000]) == want  # one intersection per edge
        assert SglSolver(s, "diamond") == want                                        # edge supports, long rows included
        assert SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x200]) == want        # ... on the oriented copy as numbered
        world = 3
        n = diamond_support_size(s, world)
        bufs = [torch.empty(n, dtype=torch.int32, device=f"cuda:{dev}") for _ in range(world)]
        for r in range(world):
            diamond_support_partial(s, bufs[r].data_ptr(), n, rank=r, world=world)
        total = torch.stack(bufs).sum(0, dtype=torch.int64)
        assert int(total.sum()) == 3 * O.tc(odag)
        tt = total.to(torch.int32)
        per = n // world
        assert sum(diamond_support_finish(s, tt[r * per:(r + 1) * per].contiguous().data_ptr(), per) for r in range(world)) == want


@pytest.mark.gpu
def test_two_stage_tables_when_forced_on_a_small_graph(dev, devopt):
    """DAG rows of 1025 .. 2048 entries: a graph with enough of them runs TWO task tables -- hosts with rows <= 1024 on the 1024-entry
    kernel, the others on the 2048-entry one (gm_launch.hip, split_stage).  The rule wants a rank's share of the second table to fill the
    chip twice, which no test-sized graph does: GM_TCT_SPLIT_ALWAYS forces it.  Triangle count, 3-motif (formula) and the diamond from
    the edge supports against the oracle, whole and as rank shares; then the same handle family without the switch (one table)."""
    g = _dense_random_graph(2000, 0.9, 77)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    dmax = int(np.diff(odag.row_ptr).max())
    assert 1024 < dmax <= 2048
    want_tc, want_dia = O.tc(odag), O.diamond(osym)
    devopt("GM_TCT_SPLIT_ALWAYS", "1")
    with g.to_device(dev) as s:
        d = s.orient()
        assert TCSolver(d) == want_tc
        assert sum(TCSolver(d, rank=r, world=3) for r in range(3)) == want_tc
        assert SglSolver(s, "diamond") == want_dia
    devopt("GM_TCT_SPLIT_ALWAYS", None)
    with g.to_device(dev) as s:
        d = s.orient()
        assert TCSolver(d) == want_tc
        assert SglSolver(s, "diamond") == want_dia


@pytest.mark.gpu
def test_key_stream_with_a_smaller_list_limit_and_without(dev, devopt):
    """the key stream of the triangle count is indexed with 32 bits: when the keys of all lists of <= 32 entries do not fit, the limit of
    a "short" list is halved until they do, and below 4 the handle goes without a stream (full task lists).  GM_KST_MAX_KEYS lowers
    the bound so that a test-sized graph takes both exits: same count as the oracle either way, and as rank shares."""
    g = rmat_csr_numpy(13, 12, seed=5)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.tc(O.orient(osym))
    with g.to_device(dev) as s:
        assert TCSolver(s.orient()) == want
    ne_dag = int(O.orient(osym).row_ptr[-1])  # (the keys of the short lists are a small multiple of the DAG's entries)
    for limit in (2 * ne_dag, ne_dag // 2, 16):  # the limit halved once or twice, more often, and no stream at all
        devopt("GM_KST_MAX_KEYS", str(max(limit, 1)))
        with g.to_device(dev) as s:
            d = s.orient()
            assert TCSolver(d) == want
            assert sum(TCSolver(d, rank=r, world=2) for r in range(2)) == want
    devopt("GM_KST_MAX_KEYS", None)


@pytest.mark.gpu
def test_edge_supports_from_the_key_stream(dev, devopt):
    """the edge supports read the short lists from the key stream (with the entries of the streamed key and of the task's own edge beside
    it) where matches are rare; GM_SUP_STREAM forces either path on a graph with many triangles per edge: diamond against the oracle and
    the per-edge kernels, after a triangle count built the stream without the entries (second set) and before one, and as rank shares."""
    import torch

    from graphminer_amd.solvers import diamond_support_finish, diamond_support_partial, diamond_support_size

    g = rmat_csr_numpy(12, 24, seed=9)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want, want_tc = O.diamond(osym), O.tc(O.orient(osym))
    for stream in ("1", "0"):
        devopt("GM_SUP_STREAM", stream)
        with g.to_device(dev) as s:
            assert SglSolver(s, "diamond") == want                   # the stream built WITH the entries (or the task lists)
            assert TCSolver(s.orient()) == want_tc
        with g.to_device(dev) as s:
            assert MotifSolver(s, 3)[1] == want_tc                   # formula 3-motif: the triangle count builds the stream of the cached DAG
            assert SglSolver(s, "diamond") == want                   # ... and the supports add the second set
            world = 3
            n = diamond_support_size(s, world)
            bufs = [torch.zeros(n, dtype=torch.int32, device=f"cuda:{dev}") for _ in range(world)]
            for r in range(world):
                diamond_support_partial(s, bufs[r].data_ptr(), n, rank=r, world=world)
            total = torch.stack(bufs).sum(0, dtype=torch.int64).to(torch.int32)
            per = n // world
            assert sum(diamond_support_finish(s, total[r * per:(r + 1) * per].contiguous().data_ptr(), per) for r in range(world)) == want
    devopt("GM_SUP_STREAM", None)


def _renumbered_expected(g, keydeg, descending):
    """numpy restatement of get_relabeled: ids by (degree, old id), rows ascending"""
    nv = g.V()
    order = np.lexsort((np.arange(nv), keydeg))
    newid = np.empty(nv, dtype=np.int64)
    newid[order] = (nv - 1 - np.arange(nv)) if descending else np.arange(nv)
    src = np.repeat(np.arange(nv), np.diff(g.row_ptr))
    keys = np.sort((newid[src] << 32) | newid[g.col_idx.astype(np.int64)])
    rp = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum(np.bincount(keys >> 32, minlength=nv), out=rp[1:])
    return rp, (keys & 0xFFFFFFFF).astype(np.int32)


def _hub_graph(seed, hub_degs, nv=9000, background=30000):
    """a sparse random symmetric graph with planted hubs of the given degrees (rows on either side of the 64-entry / LDS limits of
    the renumbering kernels)"""
    rng = np.random.default_rng(seed)
    s = [rng.integers(0, nv, background)]
    d = [rng.integers(0, nv, background)]
    for u, k in enumerate(hub_degs):
        s.append(np.full(k, u))
        d.append(rng.choice(np.arange(len(hub_degs), nv), size=k, replace=False))
    return csr_from_pairs(nv, np.concatenate(s).astype(np.uint64), np.concatenate(d).astype(np.uint64))


@pytest.mark.parametrize("path", ["lds", "global_sort"])
@pytest.mark.parametrize("graph", ["citeseer", "rmat12", "hubs_lds", "hubs_beyond_lds", "hubs_beyond_block"])
def test_renumbered_copies_are_the_permuted_graph(dev, graph, path, devopt):
    """gm_graph_renumbered (the copies the SgL / TC / k-clique kernels run on): modes 0 / 1 of the symmetric graph, mode 2 of its
    orientation, against a numpy restatement -- for the rows sorted inside the writing kernels (rank among <= 64 entries, bitonic
    network in LDS: a wave up to 1024 entries, a workgroup up to 4096; longer rows through a segmented radix sort of their
    segments) and for the device-wide radix sort of 64-bit entry keys (GM_RELABEL_GLOBAL_SORT=1). Orientation with the kept entries packed in pass 0 against the two-gather passes."""
    if path == "global_sort":
        devopt("GM_RELABEL_GLOBAL_SORT", "1")
        devopt("GM_ORIENT_TWO_GATHERS", "1")
    if graph == "citeseer":
        g = load_graph("citeseer")
    elif graph == "rmat12":
        g = rmat_csr_numpy(12, 8, 7)
    elif graph == "hubs_lds":
        g = _hub_graph(5, [63, 64, 65, 66, 127, 128, 129, 300, 1023, 1024, 1025, 2049, 4000, 4060])
    elif graph == "hubs_beyond_lds":
        g = _hub_graph(6, [64, 65, 4096, 4097, 6000])
    else:  # hubs of tens of thousands of entries through the segmented radix sort
        g = _hub_graph(8, [1000, 5000, 32768, 32769, 33000, 45000], nv=50000, background=100000)
    sdeg = np.diff(g.row_ptr)
    sym = g.to_device(dev)
    for mode in (0, 1):
        got = sym.renumbered(mode)
        rp, ci = _renumbered_expected(g, sdeg, mode == 1)
        assert np.array_equal(got.row_ptr, rp) and np.array_equal(got.col_idx, ci), (graph, mode)
    dag = sym.orient()
    want = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    host_dag = dag.download()
    assert np.array_equal(host_dag.row_ptr, want.row_ptr) and np.array_equal(host_dag.col_idx, want.col_idx)
    got = dag.renumbered(2)
    rp, ci = _renumbered_expected(host_dag, sdeg, False)
    assert np.array_equal(got.row_ptr, rp) and np.array_equal(got.col_idx, ci), (graph, 2)
    src = np.repeat(np.arange(g.V()), np.diff(got.row_ptr))
    assert np.all(got.col_idx > src)  # topological: every edge from a smaller to a larger id
    dag.free()
    sym.free()


@pytest.mark.parametrize("two_gathers", [False, True])
def test_orientation_around_the_byte_cap_of_the_degrees(dev, two_gathers, devopt):
    """the orientation passes compare degrees capped at 255 in one byte per vertex and read the exact degree only for an entry between
    two vertices of >= 255 neighbours: hubs of 253 .. 257 / 300 neighbours, ties on either side of the cap, every pair of hubs adjacent,
    against the oracle's Graph::orientation (GM_ORIENT_TWO_GATHERS=1: the exact degrees gathered in both passes)"""
    if two_gathers:
        devopt("GM_ORIENT_TWO_GATHERS", "1")
    degs = [253, 254, 254, 255, 255, 255, 256, 256, 257, 300, 300, 1500]
    nh = len(degs)
    s, d, nxt = [], [], nh
    for u, k in enumerate(degs):
        leaves = k - (nh - 1)
        s.append(np.full(leaves, u)); d.append(np.arange(nxt, nxt + leaves)); nxt += leaves
        s.append(np.full(nh - 1 - u, u)); d.append(np.arange(u + 1, nh))
    g = csr_from_pairs(nxt, np.concatenate(s).astype(np.uint64), np.concatenate(d).astype(np.uint64))
    assert sorted(np.diff(g.row_ptr)[:nh].tolist()) == sorted(degs)
    sym = g.to_device(dev)
    dag = sym.orient()
    want = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    got = dag.download()
    assert np.array_equal(got.row_ptr, want.row_ptr) and np.array_equal(got.col_idx, want.col_idx)
    assert TCSolver(dag) == math.comb(nh, 3)
    dag.free()
    sym.free()


@pytest.mark.parametrize("world,policy", [(2, 0), (3, 0), (8, 0), (4, 1)])
def test_rank_shares_of_separate_handles_add_up(dev, world, policy):
    """What a real N-GPU job does and the in-process share tests of rounds 2 - 4 did not: every rank builds ITS OWN task lists (their
    order inside a host is the order in which the placement's atomics arrived -- different on every build) and takes its share of the
    chunks.  A heavy chunk is cut into parts = every nparts-th batch of the host's task list: parts of one chunk split between two
    ranks would each cut their own order, and the shares would not add up (round 5: bench.py --gpus 2 on R-MAT-22 counted 750,563,783
    triangles for 750,506,260).  All parts of a chunk go to one rank now (ShareOrder, gm_host.h).  One handle per rank here, parts
    forced on a small graph (tune[6] & 0x1000); several builds, because an order of arrival can coincide."""
    g = rmat_csr_numpy(15, 24, seed=11)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    want_tc, want_m3 = O.tc(odag), O.motif3(osym)
    parts = [0, 0, 0, 0, 0, 0, 0x1000]
    for attempt in range(3):
        syms = [g.to_device(dev) for _ in range(world)]
        dags = [s.orient() for s in syms]
        try:
            assert sum(TCSolver(dags[r], rank=r, world=world, policy=policy, tune=parts) for r in range(world)) == want_tc, attempt
            assert sum(TCSolver(dags[r], rank=r, world=world, policy=policy) for r in range(world)) == want_tc, attempt
            got = [MotifSolver(syms[r], 3, rank=r, world=world, policy=policy, tune=parts) for r in range(world)]  # (formula: partials mod 2^64)
            assert [sum(x[i] for x in got) % 2**64 for i in range(2)] == want_m3, attempt
            assert sum(CliqueSolver(dags[r], 4, rank=r, world=world, policy=policy, tune=parts) for r in range(world)) == O.clique(odag, 4)
        finally:
            for h in dags + syms:
                h.free()


@pytest.mark.parametrize("nv", [40003, 33000, 36864])
def test_clique4_blocked_gather_with_an_unaligned_core_base(dev, nv):
    """the BLOCKED gather of the wide vertices' core rows (csrc/gm_cgather.hip cgatherb_kernel, round 6) on graphs of more than 32768 vertices
    whose size is NOT a multiple of 32: the core's base is then not word aligned (core_base & 31 = 3 / 8; 36864: aligned) and the column
    tables / block images count their bits from the base rounded down.  A dense block of hubs (DAG rows of every wide class up to ~800
    entries) + leaves hanging on them; blocked = row-major gather (tune[6] & 0x8000000) = the all-in-the-mining-kernel build = oracle,
    rank shares included, and the library confirms that the blocked path ran."""
    rng = np.random.default_rng(nv)
    nh = 1500
    iu, ju = np.triu_indices(nh, 1)
    keep = rng.random(iu.size) < 0.5
    s, d = [iu[keep]], [ju[keep]]
    leaves = np.arange(nh, nv)
    for _ in range(3):  # every leaf on three hubs: the hubs keep the highest degrees, the leaves fill the core below them
        s.append(rng.integers(0, nh, leaves.size))
        d.append(leaves)
    g = csr_from_pairs(nv, np.concatenate(s).astype(np.uint64), np.concatenate(d).astype(np.uint64))
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    assert int(np.diff(odag.row_ptr).max()) > 512
    want = O.clique(odag, 4)
    with g.to_device(dev) as sym:
        dag = sym.orient()
        got = CliqueSolver(dag, 4)
        info = (C.c_int64 * 4)()
        _lib.check(_lib.load().gm_clique4_gather_info(dag.handle, info), "gm_clique4_gather_info")
        assert info[0] > 0 and info[2] > 0 and info[3] > 1, list(info)  # units, work items, blocks: the blocked gather ran
        assert got == want
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x8000000]) == want  # the row-major gather of round 5
        assert CliqueSolver(dag, 4, tune=[0, 0, 0, 0, 0, 0, 0x40000]) == want
        assert sum(CliqueSolver(dag, 4, rank=r, world=3) for r in range(3)) == want
        assert sum(CliqueSolver(dag, 4, rank=r, world=2, policy=1, tune=[0, 0, 0, 0, 0, 0, 0x8000000]) for r in range(2)) == want
        dag.free()
