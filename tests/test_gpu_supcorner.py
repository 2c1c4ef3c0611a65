"""Edge supports: the supports of the HUB CORNER's edges on the matrix cores (gm_ctc.hip, core_tc_block_kernel<true>) -- the task lists leave
the rows of the last H vertices out, t(i, j) = (A A)_ij over the symmetric corner A is added to the support array entry by entry, and the
diamond count sum C(t, 2) must equal the oracle's (src/sgl/cpu_kernels/diamond.h:1-14) -- sensitive to every single support.

GM_SUP_CORE_H / GM_TOPO_MIN_ROW are read when a handle's renumbered copy and task lists are built: every case uploads a fresh graph."""
import numpy as np
import pytest

import oracle as O
from graphminer_amd import SglSolver, TCSolver
from graphminer_amd.rmat import csr_from_pairs, rmat_csr_numpy

pytestmark = pytest.mark.gpu
NO_MASKS = [0, 0, 0, 0, 0, 0, 0x40000000]
PER_EDGE = [0, 0, 0, 0, 0, 0, 0x10000000]
NO_STREAM = [0, 0, 0, 0, 0, 0, 0x20000000]


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available()
    return 0


def _supports(s, dev, world=1):
    import torch

    from graphminer_amd.solvers import diamond_support_partial, diamond_support_size

    n = diamond_support_size(s, world)
    bufs = [torch.full((n,), 5, dtype=torch.int32, device=f"cuda:{dev}") for _ in range(world)]
    for r in range(world):
        diamond_support_partial(s, bufs[r].data_ptr(), n, rank=r, world=world)
    return torch.stack(bufs).sum(0, dtype=torch.int64)


@pytest.mark.parametrize("h", [0, 512, 1024, 4096, 1 << 14])
def test_diamond_with_every_corner_size_equals_the_oracle(dev, h, devopt):
    devopt("GM_TOPO_MIN_ROW", "0")
    devopt("GM_SUP_CORE_H", str(h))
    devopt("GM_SUP_STREAM", "0")
    g = rmat_csr_numpy(14, 24, seed=3 + h)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    want, want_tc = O.diamond(osym), O.tc(odag)
    with g.to_device(dev) as s:
        assert SglSolver(s, "diamond") == want
        assert SglSolver(s, "diamond") == want  # again: every launch zeroes and refills the supports
        assert SglSolver(s, "diamond", tune=NO_MASKS) == want
        assert SglSolver(s, "diamond", tune=PER_EDGE) == want
        one = _supports(s, dev, 1)
        assert int(one.sum()) == 3 * want_tc  # three increments per triangle
        for world in (2, 3):  # a rank's share of the chunks and every world-th block of the corner into its own array
            assert bool((_supports(s, dev, world)[: one.numel()] == one).all()), world
        with s.orient() as dag:  # the triangle count on the same kind of handle, with and without its key stream
            assert TCSolver(dag) == want_tc
            assert TCSolver(dag, tune=NO_STREAM) == want_tc
            assert sum(TCSolver(dag, tune=NO_STREAM, rank=r, world=3) for r in range(3)) == want_tc


def test_diamond_corner_on_a_dense_block_with_leaves(dev, devopt):
    """600 hubs, nearly complete among themselves, and leaves that give them strictly ascending degrees: every block of the corner's product
    is nearly full and the diagonal blocks matter (edges i < j only)"""
    devopt("GM_TOPO_MIN_ROW", "0")
    devopt("GM_SUP_CORE_H", "1024")
    rng = np.random.default_rng(5)
    n = 600
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random(iu.size) < 0.9
    s_, d_ = [iu[keep]], [ju[keep]]
    nxt = n
    for v in range(n):
        s_.append(np.full(v + 1, v))
        d_.append(np.arange(nxt, nxt + v + 1))
        nxt += v + 1
    g = csr_from_pairs(nxt, np.concatenate(s_).astype(np.uint64), np.concatenate(d_).astype(np.uint64))
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.diamond(osym)
    with g.to_device(dev) as s:
        assert SglSolver(s, "diamond") == want
        assert SglSolver(s, "diamond", tune=PER_EDGE) == want


def test_default_rule_on_rmat20(dev):
    """no switches: the density rule picks a corner on R-MAT-20; diamond against the per-edge kernels and the oracle"""
    g = rmat_csr_numpy(20, 8, seed=42)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.diamond(osym)
    with g.to_device(dev) as s:
        assert SglSolver(s, "diamond") == want
        assert SglSolver(s, "diamond", tune=PER_EDGE) == want
