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


def test_house_pentagon(graphs):
    name, _, sym, _ = graphs
    e = GOLDEN[name]
    if "house" not in e or e["ne"] > 5000 and e["kind"] == "rmat":
        pytest.skip("no golden / too slow")
    assert O.house(sym) == e["house"]
    assert O.pentagon(sym) == e["pentagon"]


@pytest.mark.parametrize("k", [4, 5, 6, 7])
def test_clique(graphs, k):
    name, _, _, dag = graphs
    e = GOLDEN[name]
    if f"clique{k}" not in e:
        pytest.skip("no golden")
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
    assert c["motif4"] == [222630, 111153, 22900, 3094, 2200, 255]
