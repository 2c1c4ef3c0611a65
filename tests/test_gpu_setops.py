"""Per-primitive parity: the wave64 set operations (gm_setop_batch) against the CPU oracle's
merge-based set operations on adversarial sorted lists."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from graphminer_amd import _lib

pytestmark = pytest.mark.gpu


def _lists():
    rng = np.random.default_rng(0)
    L = [np.array([], dtype=np.int32), np.array([5], dtype=np.int32), np.array([0], dtype=np.int32)]
    for n in (2, 31, 63, 64, 65, 127, 128, 129, 1000, 4097):
        L.append(np.sort(rng.choice(4 * n + 8, n, replace=False)).astype(np.int32))
        L.append(np.arange(n, dtype=np.int32))            # dense prefix, "all-equal prefixes" with its twin
        L.append(np.arange(n, dtype=np.int32) * 2)        # evens
        L.append(np.arange(n, dtype=np.int32) * 2 + 1)    # odds: disjoint from evens
    L.append(np.array([2**31 - 2], dtype=np.int32))
    return L


def _pairs():
    L = _lists()
    pairs = [(a, b) for a in L for b in L]
    rng = np.random.default_rng(1)
    idx = rng.permutation(len(pairs))[:900]
    must = [(L[0], L[0]), (L[0], L[5]), (L[5], L[0]), (L[1], L[1])]
    return must + [pairs[i] for i in idx]


@pytest.fixture(scope="module")
def batch():
    import torch

    assert torch.cuda.is_available()
    pairs = _pairs()
    vals, ab, ae, bb, be = [], [], [], [], []
    pos = 0
    for a, b in pairs:
        ab.append(pos); pos += a.size; ae.append(pos)
        bb.append(pos); pos += b.size; be.append(pos)
        vals += [a, b]
    vals = np.concatenate(vals).astype(np.int32)
    rng = np.random.default_rng(2)
    upper = []
    for a, b in pairs:  # bounds below / inside / above all keys
        hi = int(max(a.max() if a.size else 0, b.max() if b.size else 0))
        upper.append(int(rng.choice([0, 1, hi // 2, hi, min(hi + 1, 2**31 - 1), 2**31 - 1])))
    skip = [int(a[rng.integers(a.size)]) if a.size and i % 2 else -1 for i, (a, b) in enumerate(pairs)]
    dev = torch.device("cuda", 0)
    t = lambda x, dt: torch.tensor(np.asarray(x), dtype=dt, device=dev)
    return dict(pairs=pairs, vals=t(vals, torch.int32), ab=t(ab, torch.int64), ae=t(ae, torch.int64), bb=t(bb, torch.int64),
                be=t(be, torch.int64), upper=t(upper, torch.int32), skip=t(skip, torch.int32), h_upper=upper, h_skip=skip,
                h_ab=ab, n=len(pairs), dev=dev)


def _run(op, B, with_upper, with_skip, is_set):
    import torch

    lib = _lib.load()
    out_num = torch.zeros(B["n"], dtype=torch.int32, device=B["dev"])
    out_vals = torch.full_like(B["vals"], -7) if is_set else None
    rc = lib.gm_setop_batch(op, B["n"], B["vals"].data_ptr(), B["ab"].data_ptr(), B["ae"].data_ptr(), B["bb"].data_ptr(),
                            B["be"].data_ptr(), B["upper"].data_ptr() if with_upper else None,
                            B["skip"].data_ptr() if with_skip else None, out_num.data_ptr(),
                            out_vals.data_ptr() if is_set else None, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out_num.cpu().numpy().astype(np.int64), (out_vals.cpu().numpy() if is_set else None)


def _p(x):
    return x.ctypes.data if x.size else None


def test_intersect_num(batch):
    got, _ = _run(_lib.GM_OP_INTERSECT_NUM, batch, False, False, False)
    L = O.lib()
    for i, (a, b) in enumerate(batch["pairs"]):
        assert got[i] == L.gmo_intersect_num(_p(a), a.size, _p(b), b.size), i


def test_intersect_num_upper(batch):
    got, _ = _run(_lib.GM_OP_INTERSECT_NUM_UPPER, batch, True, False, False)
    L = O.lib()
    for i, (a, b) in enumerate(batch["pairs"]):
        assert got[i] == L.gmo_intersect_num_upper(_p(a), a.size, _p(b), b.size, batch["h_upper"][i]), i


@pytest.mark.parametrize("bounded", [False, True])
def test_intersect_set(batch, bounded):
    op = _lib.GM_OP_INTERSECT_SET_UPPER if bounded else _lib.GM_OP_INTERSECT_SET
    num, vals = _run(op, batch, bounded, False, True)
    L = O.lib()
    for i, (a, b) in enumerate(batch["pairs"]):
        out = np.zeros(max(a.size, b.size, 1), dtype=np.int32)
        if bounded:
            n = L.gmo_intersect_set_upper(_p(a), a.size, _p(b), b.size, batch["h_upper"][i], out.ctypes.data)
        else:
            n = L.gmo_intersect_set(_p(a), a.size, _p(b), b.size, out.ctypes.data)
        assert num[i] == n, i
        s = batch["h_ab"][i]
        assert np.array_equal(vals[s:s + n], out[:n]), i  # ascending, same order as the merge


@pytest.mark.parametrize("bounded", [False, True])
def test_difference(batch, bounded):
    """A \\ B with the CPU quirk 'also drop other.vid' (src/common/VertexSet.cc:29,37) through skip."""
    L = O.lib()
    num, _ = _run(_lib.GM_OP_DIFFERENCE_NUM_UPPER if bounded else _lib.GM_OP_DIFFERENCE_NUM, batch, bounded, True, False)
    snum, vals = _run(_lib.GM_OP_DIFFERENCE_SET_UPPER if bounded else _lib.GM_OP_DIFFERENCE_SET, batch, bounded, True, True)
    for i, (a, b) in enumerate(batch["pairs"]):
        out = np.zeros(max(a.size, 1), dtype=np.int32)
        sk = batch["h_skip"][i]
        if bounded:
            up = batch["h_upper"][i]
            n = L.gmo_difference_set_upper(_p(a), a.size, _p(b), b.size, sk, up, out.ctypes.data)
            assert n == L.gmo_difference_num_upper(_p(a), a.size, _p(b), b.size, sk, up)
        else:
            n = L.gmo_difference_set(_p(a), a.size, _p(b), b.size, sk, out.ctypes.data)
        assert num[i] == n and snum[i] == n, i
        s = batch["h_ab"][i]
        assert np.array_equal(vals[s:s + n], out[:n]), i


def test_count_smaller(batch):
    """count_smaller (include/operations.cuh:61-105; the listing form of diamond): |{x in A : x < bound}| = the oracle's
    bounded() prefix length (include/VertexSet.h:240-255)"""
    got, _ = _run(_lib.GM_OP_COUNT_SMALLER, batch, True, False, False)
    L = O.lib()
    for i, (a, b) in enumerate(batch["pairs"]):
        assert got[i] == L.gmo_bounded(_p(a), a.size, batch["h_upper"][i]) == int(np.searchsorted(a, batch["h_upper"][i], side="left")), i
