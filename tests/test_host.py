"""CPU-side tests of the host logic and of the C-ABI surface (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import GOLDEN, csr_sha, load_graph
from graphminer_amd import _lib
from graphminer_amd.graph import Graph, GraphFormatError
from graphminer_amd.rmat import rmat_csr_numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "graphminer_amd.h")).read()
    declared = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", header))
    declared -= {"gm_status"}
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.gm_version() >= 100
    assert lib.gm_strerror(_lib.GM_ERR_UNSUPPORTED) == b"Not implemented"


def test_no_device_is_an_error_not_a_fallback():
    """On a box without a GPU every compute entry point must fail loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    n = C.c_int(-1)
    assert lib.gm_device_count(C.byref(n)) != _lib.GM_OK and n.value == 0
    g = load_graph("citeseer")
    with pytest.raises(_lib.GraphMinerError):
        g.to_device(0)


def test_argument_validation_without_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.gm_graph_upload(None, 0, C.byref(h)) == _lib.GM_ERR_INVALID
    tot = C.c_uint64(7)
    assert lib.gm_tc(None, None, C.byref(tot), None) == _lib.GM_ERR_INVALID
    assert lib.gm_setop_batch(99, 1, None, None, None, None, None, None, None, None, None, None) == _lib.GM_ERR_INVALID
    assert lib.gm_rmat_keys(0, 10, 1, None, None) == _lib.GM_ERR_INVALID


def test_loader_reads_fixture_and_matches_reference_meta():
    g = Graph(os.path.join(ROOT, "tests", "fixtures", "citeseer", "graph"))
    assert (g.V(), g.E(), g.get_max_degree()) == (3312, 9072, 99)
    assert g.name == "citeseer"
    assert g.print_meta_data() == "|V|: 3312, |E|: 9072, Max Degree: 99"
    # rows strictly ascending, symmetric, no self loops
    for v in (0, 17, 3311):
        r = g.N(v)
        assert np.all(np.diff(r) > 0) and v not in r
        for u in r:
            assert v in g.N(int(u))


def test_loader_error_behaviour(tmp_path):
    with pytest.raises(FileNotFoundError):
        Graph(str(tmp_path / "nope" / "graph"))
    g = rmat_csr_numpy(6, 4, 1)
    p = str(tmp_path / "g" / "graph")
    g.save(p)
    assert csr_sha(Graph(p)) == csr_sha(g)
    meta = open(p + ".meta.txt").read().split()
    meta[2] = "8"  # sizeof(vidType) != 4 -> graph.cc:30 assert
    open(p + ".meta.txt", "w").write("\n".join(meta))
    with pytest.raises(GraphFormatError):
        Graph(p)
    meta[2], meta[6] = "4", str(g.V())  # max_degree < nv violated -> graph.cc:34
    open(p + ".meta.txt", "w").write("\n".join(meta))
    with pytest.raises(GraphFormatError):
        Graph(p)


def test_rmat_is_deterministic_and_pinned():
    for name, e in GOLDEN.items():
        if name.startswith("_") or e["kind"] != "rmat" or e["scale"] > 12:
            continue
        g = rmat_csr_numpy(e["scale"], e["edge_factor"], e["seed"])
        assert csr_sha(g) == e["csr_sha256"], name
        assert g.max_degree == e["max_degree"]
    a, b = rmat_csr_numpy(8, 4, 1), rmat_csr_numpy(8, 4, 2)
    assert csr_sha(a) != csr_sha(b)


def test_empty_and_ragged_host_graphs():
    g = Graph(row_ptr=[0, 0, 0], col_idx=[])
    assert (g.V(), g.E(), g.max_degree) == (2, 0, 0)
    with pytest.raises(GraphFormatError):
        Graph(row_ptr=[0, 2, 1], col_idx=[1])


def test_size_limits_are_reported_not_wrapped():
    """maximum sizes: the mining kernels index tasks with int32. A graph of 2^31 entries or more uploads as a BIG handle (64-bit
    offsets: orientation, formula 3-motif, download -- tests/test_gpu_parity.py); beyond 2^40 entries or 2^31 - 2 vertices it is
    refused, never truncated (without a device the upload of an acceptable size fails with NO_DEVICE instead)"""
    import torch

    lib = _lib.load()
    dummy = np.zeros(4, dtype=np.int64)
    h = C.c_void_p()
    big = _lib.gm_csr(10, 2**40, 5, dummy.ctypes.data, dummy.ctypes.data)
    assert lib.gm_graph_upload(C.byref(big), 0, C.byref(h)) == _lib.GM_ERR_TOO_LARGE
    many = _lib.gm_csr(2**31 - 2, 4, 5, dummy.ctypes.data, dummy.ctypes.data)
    assert lib.gm_graph_upload(C.byref(many), 0, C.byref(h)) == _lib.GM_ERR_TOO_LARGE
    if not torch.cuda.is_available():
        ok = _lib.gm_csr(10, 2**31, 5, dummy.ctypes.data, dummy.ctypes.data)
        assert lib.gm_graph_upload(C.byref(ok), 0, C.byref(h)) == _lib.GM_ERR_NO_DEVICE
    neg = _lib.gm_csr(-1, 0, 0, dummy.ctypes.data, dummy.ctypes.data)
    assert lib.gm_graph_upload(C.byref(neg), 0, C.byref(h)) == _lib.GM_ERR_INVALID
    assert lib.gm_strerror(_lib.GM_ERR_TOO_LARGE).startswith(b"graph exceeds")


def test_partition_and_chunk_table_host_api():
    from graphminer_amd import dist

    g = load_graph("cora")
    t = dist.chunk_table(g.row_ptr)
    assert t[:, 3].max() == g.E() and (t[:, 1] > t[:, 0]).all()
    f, s_, c = C.c_int64(), C.c_int64(), C.c_int64()
    lib = _lib.load()
    assert lib.gm_partition(10, 3, 3, 0, C.byref(f), C.byref(s_), C.byref(c)) == _lib.GM_ERR_INVALID  # rank >= world
    assert lib.gm_partition(10, 2, 3, 0, C.byref(f), C.byref(s_), C.byref(c)) == 0 and (f.value, s_.value, c.value) == (2, 3, 3)
    raw = (C.c_uint64 * 6)(1381580, 123529, 54600, 7460, 6059, 255)  # citeseer raw sums (4-motif formula)
    out = (C.c_uint64 * 6)()
    assert lib.gm_motif4_finish(raw, out) == 0
    assert [int(x) for x in out] == GOLDEN["citeseer"]["motif4"]


# ---- edgelist2bin: text edge lists -> the three-file format (reference README.md:104 points to an external converter) ----
E2B = os.path.join(ROOT, "graphminer_amd", "bin", "edgelist2bin")


def _e2b(*args):
    import subprocess

    return subprocess.run([E2B, *map(str, args)], capture_output=True, text=True)


@pytest.mark.skipif(not os.path.exists(E2B), reason="edgelist2bin not built (make -C graphminer_amd tools)")
@pytest.mark.parametrize("name", ["citeseer", "cora"])
def test_edgelist2bin_round_trip_is_byte_identical(name, tmp_path):
    """fixture -> text dump (every undirected edge once) -> convert: vertex.bin / edge.bin byte-identical, meta.txt equal"""
    src = os.path.join(ROOT, "tests", "fixtures", name, "graph")
    meta = open(src + ".meta.txt").read().split()
    txt, out = tmp_path / "edges.txt", tmp_path / "graph"
    assert _e2b("--dump", src, txt).returncode == 0
    r = _e2b("--meta-tail", " ".join(meta[7:10]), txt, out)
    assert r.returncode == 0, r.stderr
    for ext in (".vertex.bin", ".edge.bin"):
        assert open(str(out) + ext, "rb").read() == open(src + ext, "rb").read(), ext
    assert open(str(out) + ".meta.txt").read().split() == meta[:10]  # (cora's meta.txt goes on with mask ranges the loader never reads)
    g = Graph(str(out))  # and the loader (with the reference's asserts) accepts it
    assert f"|V|: {g.V()}, |E|: {g.E()}, Max Degree: {g.max_degree}" in r.stdout


@pytest.mark.skipif(not os.path.exists(E2B), reason="edgelist2bin not built (make -C graphminer_amd tools)")
def test_edgelist2bin_symmetrises_dedupes_and_renumbers(tmp_path):
    """directed duplicates, both directions, self loops, comments, 1-based sparse ids, commas and tabs -> csr_from_pairs' graph"""
    from graphminer_amd.rmat import csr_from_pairs

    rng = np.random.default_rng(5)
    ids = np.sort(rng.choice(np.arange(1, 5000), size=300, replace=False))  # sparse 1-based id space
    s, d = rng.integers(0, 300, 4000), rng.integers(0, 300, 4000)
    txt = tmp_path / "g.txt"
    with open(txt, "w") as f:
        f.write("# Directed graph (each unordered pair of nodes is saved once or more)\n# FromNodeId\tToNodeId\n")
        for i, (a, b) in enumerate(zip(s, d)):
            sep = ["\t", " ", ",", "  "][i % 4]
            f.write(f"{ids[a]}{sep}{ids[b]}\n")
            if i % 7 == 0:
                f.write(f"{ids[b]} {ids[a]}\n")  # the reverse direction as well
        f.write("\n")
    r = _e2b("--one-based", "--compact", txt, tmp_path / "c")
    assert r.returncode == 0, r.stderr
    want = csr_from_pairs(300, s.astype(np.uint64), d.astype(np.uint64))
    got = Graph(str(tmp_path / "c"))
    # (--compact drops vertices without an edge; the random pairs touch all 300 with overwhelming probability)
    assert got.V() == want.V() and np.array_equal(got.row_ptr, want.row_ptr) and np.array_equal(got.col_idx, want.col_idx)
    assert got.max_degree == want.max_degree
    # without --compact: ids kept (minus one), nv = largest id
    r = _e2b("--one-based", txt, tmp_path / "k")
    assert r.returncode == 0, r.stderr
    keep = Graph(str(tmp_path / "k"))
    assert keep.V() == int(ids[np.union1d(s, d)].max()) and keep.E() == want.E()
    deg = np.diff(keep.row_ptr)
    assert np.array_equal(deg[ids[np.arange(300)] - 1], np.diff(want.row_ptr))


@pytest.mark.skipif(not os.path.exists(E2B), reason="edgelist2bin not built (make -C graphminer_amd tools)")
def test_edgelist2bin_matrix_market_and_errors(tmp_path):
    mtx = tmp_path / "m.mtx"
    mtx.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n% comment\n4 4 4\n2 1\n3 1 1.0\n4 3\n3 3\n")
    assert _e2b(mtx, tmp_path / "m").returncode == 0
    g = Graph(str(tmp_path / "m"))
    assert (g.V(), g.E(), g.max_degree) == (4, 6, 2)
    assert g.col_idx.tolist() == [1, 2, 0, 0, 3, 2]
    empty = tmp_path / "e.txt"
    empty.write_text("# nothing\n5 5\n")
    r = _e2b(empty, tmp_path / "e")
    assert r.returncode != 0 and "no edge" in r.stderr  # the loader asserts 0 < max_degree < nv (graph.cc:34)
    bad = tmp_path / "b.txt"
    bad.write_text("1 x\n")
    assert _e2b(bad, tmp_path / "b").returncode != 0
