"""The CPU oracle (oracle/gm_oracle.c) against the golden vectors produced by the REAL reference
binaries (tests/golden/make_golden.py) -- this is what pins the oracle."""
import pytest

import oracle as O
from common import GOLDEN, GRAPH_NAMES, csr_sha, load_graph


@pytest.fixture(scope="module", params=GRAPH_NAMES)
def graphs(request):
    name = request.param
    g = load_graph(name)
    sym = O.OGraph(g.row_ptr, g.col_idx)
    return name, g, sym, O.orient(sym)


def test_inputs_are_the_golden_inputs(graphs):
    name, g, _, _ = graphs
    e = GOLDEN[name]
    assert (g.V(), g.E(), g.max_degree) == (e["nv"], e["ne"], e["max_degree"])
    assert csr_sha(g) == e["csr_sha256"]


def test_orientation_matches_reference_meta(graphs):
    name, _, _, dag = graphs
    e = GOLDEN[name]
    assert dag.ne == e["dag_ne"]            # |E| printed after Graph::orientation (graph.cc:646)
    assert dag.c.max_degree == e["dag_max_degree"]


def test_tc(graphs):
    name, _, _, dag = graphs
    assert O.tc(dag) == GOLDEN[name]["tc"]
    assert O.clique(dag, 3) == GOLDEN[name]["tc"]


def test_diamond_rectangle(graphs):
    name, _, sym, _ = graphs
    if GOLDEN[name]["ne"] > 100000:
        pytest.skip("pair enumeration too slow for the CPU suite at this size (covered up to rmat12)")
    assert O.diamond(sym) == GOLDEN[name]["diamond"]
    assert O.rectangle(sym) == GOLDEN[name]["rectangle"]


def test_tailedtriangle_4path_3star(graphs):
    """the other 4-vertex SgL patterns (src/sgl/cpu_kernels/tailedtriangle.h, 4path.h, 3star.h): oracle = sgl_omp_base on every golden graph,
    = the closed forms the HIP path uses (sums of the per-edge quantities of the formula 4-motif, csrc/gm_launch.hip gm_sgl)"""
    name, g, sym, _ = graphs
    e = GOLDEN[name]
    if "3star" not in e:
        pytest.skip("no golden")
    assert (O.star3(sym), O.path4(sym), O.tailedtriangle(sym)) == (e["3star"], e["4path"], e["tailedtriangle"])
    if "motif4" in e:  # the identities with the vertex-induced 4-motif counts [3-star, 4-path, tailed triangle, 4-cycle, diamond, 4-clique]
        m = e["motif4"]
        assert e["3star"] == m[0] + m[2] + 2 * m[4] + 4 * m[5]
        assert e["tailedtriangle"] == m[2] + 4 * m[4] + 12 * m[5]
        assert e["4path"] == m[1] + 2 * m[2] + 4 * m[3] + 6 * m[4] + 12 * m[5]


def test_house_pentagon(graphs):
    name, _, sym, _ = graphs
    e = GOLDEN[name]
    if "house" not in e or e["ne"] > 5000 and e["kind"] == "rmat":
        pytest.skip("no golden / too slow")
    assert O.house(sym) == e["house"]
    assert O.pentagon(sym) == e["pentagon"]


@pytest.mark.parametrize("k", [4, 5, 6, 7, 8, 9, 10, 11, 12])  # (9..12: goldens from the generic clique_omp_recursive)
def test_clique(graphs, k):
    name, _, _, dag = graphs
    e = GOLDEN[name]
    if f"clique{k}" not in e:
        pytest.skip("no golden")
    if e[f"clique{k}"] > 10**11:
        pytest.skip("the oracle's plain DFS needs minutes here; the GPU test checks this golden directly")
    assert O.clique(dag, k) == e[f"clique{k}"]
    if k == 4:
        assert e["kcl4"] == e["clique4"]  # Pangolin kcl_omp_base agrees (second count oracle)


def test_motif3(graphs):
    name, _, sym, _ = graphs
    assert O.motif3(sym) == GOLDEN[name]["motif3"]
    assert GOLDEN[name]["motif3"][1] == GOLDEN[name]["tc"]


def test_motif4(graphs):
    name, _, sym, _ = graphs
    e = GOLDEN[name]
    if "motif4" not in e:
        pytest.skip("no golden")
    m4 = O.motif4(sym)
    assert m4 == e["motif4"]
    # identities SURVEY 8c: edge-induced diamond = vi-diamond + 6*K4 ; 4-cycle = vi-4cycle + vi-diamond + 3*K4
    assert e["diamond"] == m4[4] + 6 * m4[5]
    assert e["rectangle"] == m4[3] + m4[4] + 3 * m4[5]


def test_readme_known_answers_small():
    # src/triangle/README.md:53, src/sgl/README.md:53, src/clique/README.md:53, src/motif/README.md:52
    c = GOLDEN["citeseer"]
    assert (c["tc"], c["diamond"], c["rectangle"], c["house"], c["pentagon"]) == (1166, 3730, 6059, 55359, 28394)
    assert (c["tailedtriangle"], c["4path"], c["3star"]) == (34760, 185589, 250950)  # sgl_omp_base on citeseer
    assert c["motif4"] == [222630, 111153, 22900, 3094, 2200, 255]


def test_strided_samples_partition_the_whole_count():
    """gmo_*_sample (bench.py's bounded CPU baselines): the samples of all offsets add up to the golden counts and tasks"""
    for name in ("cora", "rmat10_ef16_s42"):
        g = load_graph(name)
        s = O.OGraph(g.row_ptr, g.col_idx)
        d = O.orient(s)
        e = GOLDEN[name]
        for stride in (1, 3):
            ds = [O.diamond_sample(s, stride, o) for o in range(stride)]
            cs = [O.clique_sample(d, 4, stride, o) for o in range(stride)]
            ms = [O.motif3_sample(s, stride, o) for o in range(stride)]
            ts = [O.tc_sample(d, stride, o) for o in range(stride)]
            assert (sum(x[0] for x in ds), sum(x[1] for x in ds)) == (e["diamond"], e["ne"] // 2)
            assert (sum(x[0] for x in cs), sum(x[1] for x in cs)) == (e["clique4"], e["dag_ne"])
            assert (sum(x[0] for x in ts), sum(x[1] for x in ts)) == (e["tc"], e["dag_ne"])
            assert [sum(x[0][i] for x in ms) for i in range(2)] == e["motif3"] and sum(x[1] for x in ms) == e["ne"]
