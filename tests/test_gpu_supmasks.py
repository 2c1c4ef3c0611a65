"""Edge supports with MATCH MASKS (gm_sup.hip, round 5): the in-edge tasks of the triangle pass report their streamed edges as a bit
mask of their tail (plain stores) and sup_cols_kernel sums the masks of a row by column, instead of one memory-side atomic per match.
Diamond = sum C(t_e, 2) is sensitive to every single support, so equality with the oracle / the atomics path checks the whole array.

The developer option GM_SUP_MASK_MIN (the shortest tail that gets a mask) is read when a handle's masks are laid out: the variants run
on fresh handles, and through the CLI binary (`--dev NAME=VALUE`)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle as O
from common import ROOT
from graphminer_amd import SglSolver, TCSolver, _lib
from graphminer_amd.rmat import csr_from_pairs, rmat_csr_numpy

pytestmark = pytest.mark.gpu
NO_MASKS = [0, 0, 0, 0, 0, 0, 0x40000000]
PER_EDGE = [0, 0, 0, 0, 0, 0, 0x10000000]


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available()
    return 0


def cli_diamond(prefix, **opts):
    """sgl_gpu_base <graph> diamond with developer options handed over as `--dev NAME=VALUE` (the apps read nothing from the environment)"""
    exe = os.path.join(ROOT, "graphminer_amd", "bin", "sgl_gpu_base")
    dev = [a for k, v in opts.items() for a in ("--dev", f"{k}={v}")]
    r = subprocess.run([exe, prefix, "diamond", *dev], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return int(re.search(r"total_num = (\d+)", r.stdout).group(1))


@pytest.mark.parametrize("scale,ef,seed", [(12, 24, 9), (14, 16, 42), (13, 64, 5)])
def test_masks_equal_atomics_and_oracle_on_rmat(dev, scale, ef, seed, devopt, tmp_path):
    """dense R-MAT graphs (tails of every length up to several hundred keys: the flattened pass AND the long lists), task-list path forced"""
    devopt("GM_SUP_STREAM", "0")
    g = rmat_csr_numpy(scale, ef, seed=seed)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.diamond(osym)
    odag = O.orient(osym)
    assert int(np.diff(odag.row_ptr).max()) > 64, "the graph must have rows with masked tails"
    with g.to_device(dev) as s:
        got, st = SglSolver(s, "diamond", return_stats=True)
        assert got == want
        assert SglSolver(s, "diamond", tune=NO_MASKS) == want        # every streamed edge by an atomic (round 4's kernel)
        assert SglSolver(s, "diamond") == want                       # again: the arena is rewritten by every launch
        assert SglSolver(s, "diamond", tune=PER_EDGE) == want
        assert SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x800000]) == want  # the set's global fallback lookups under the masks
        assert SglSolver(s, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x1000]) == want    # heavy chunks cut into parts
        assert SglSolver(s, "diamond", chunk=64) == want
        assert TCSolver(s.orient()) == O.tc(odag)
    # the shortest masked tail: 1 (every in-edge task), 7, the default, 192 (only the long lists), 5000 (none)
    g.save(str(tmp_path / "graph"))
    masked = False
    for lmin in ("1", "7", "192", "5000"):
        assert cli_diamond(str(tmp_path / "graph"), GM_SUP_MASK_MIN=lmin, GM_SUP_STREAM="0") == want, lmin
        devopt("GM_SUP_MASK_MIN", lmin)  # (read when a handle's masks are laid out: a fresh handle per value)
        with g.to_device(dev) as s:
            got = SglSolver(s, "diamond")
            info = (C.c_int64 * 4)()
            _lib.check(_lib.load().gm_diamond_support_info(s.handle, info), "gm_diamond_support_info")
            assert got == want, (lmin, got)
            # (info[0] = 0: no masks on this graph at all -- its oriented copy is not renumbered -- and the option has nothing to shape)
            assert info[0] == 0 or info[3] == int(lmin), (lmin, list(info))
            masked = masked or info[0] > 0
    devopt("GM_SUP_MASK_MIN", None)
    assert masked or (scale, ef) != (13, 64), "the dense graph must run with match masks"
    assert cli_diamond(str(tmp_path / "graph"), GM_SUP_NO_MASKS="1", GM_SUP_STREAM="0") == want


def _clique_with_leaves(n, p, seed, base=0):
    """n hubs, every pair an edge with probability p, plus leaves so that hub i has a strictly growing degree: the DAG rows of the hubs
    are [n - 1, n - 2, ..., 0] entries long and their tails cover every length"""
    rng = np.random.default_rng(seed)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random(iu.size) < p
    s, d = [iu[keep] + base], [ju[keep] + base]
    deg = np.bincount(np.concatenate([iu[keep], ju[keep]]), minlength=n)
    nxt = base + n
    for v in range(n):  # degrees strictly ascending in v: orientation keeps i -> j for i < j
        need = int(deg.max() + 1 + v - deg[v])
        s.append(np.full(need, v + base))
        d.append(np.arange(nxt, nxt + need))
        nxt += need
    return np.concatenate(s).astype(np.uint64), np.concatenate(d).astype(np.uint64), nxt


@pytest.mark.parametrize("n,p", [(40, 0.9), (200, 0.5), (700, 0.3), (1100, 0.25), (2048, 0.1)])
def test_masks_on_planted_dense_blocks(dev, n, p, devopt):
    """one dense block of n hubs: DAG rows up to n - 1 entries (both stages, rows of exactly 2048 entries included), tails of every
    length, in-edge and out-edge tasks mixed by the host rule"""
    devopt("GM_SUP_STREAM", "0")
    s, d, nv = _clique_with_leaves(n + (1 if n == 2048 else 0), p, seed=n)
    g = csr_from_pairs(nv, s, d)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.diamond(osym)
    with g.to_device(dev) as sym:
        assert SglSolver(sym, "diamond") == want
        assert SglSolver(sym, "diamond", tune=NO_MASKS) == want
        assert SglSolver(sym, "diamond", tune=[0, 0, 0, 0, 0, 0, 0x800000]) == want


def test_masks_with_surplus_list_matches(dev, devopt):
    """ids that collide in the hashed position set (inverse of the hash multiplier): the bucket overflows, keys are found through the
    surplus list (hit1) -- their bits join the masks of the tasks that found them"""
    devopt("GM_SUP_STREAM", "0")
    inv = pow(0x9E3779B1, -1, 1 << 32)
    # ids whose hash has the same top bits: x = inv * (b << 22 | j) mod 2^32, kept below 2^22
    cand = [(inv * ((5 << 22) | j)) & 0xFFFFFFFF for j in range(1, 1 << 16)]
    ids = sorted(x for x in cand if 3000 < x < (1 << 22))[:48]
    assert len(ids) >= 24
    hubs = np.array(ids, dtype=np.int64)
    n = hubs.size
    rng = np.random.default_rng(3)
    iu, ju = np.triu_indices(n, 1)
    keep = rng.random(iu.size) < 0.8
    s, d = [hubs[iu[keep]]], [hubs[ju[keep]]]
    # vertex 0 .. 9: rows that contain all the colliding hubs (their hashed set overflows the four-slot bucket)
    for u in range(10):
        s.append(np.full(n, u))
        d.append(hubs)
    deg = np.bincount(np.concatenate(s + d), minlength=int(hubs.max()) + 1)
    nxt = int(hubs.max()) + 1
    for k, v in enumerate(hubs):  # hub degrees ascending with the id and above the rows 0 .. 9
        need = int(deg.max() + 1 + k - deg[v])
        s.append(np.full(need, v))
        d.append(np.arange(nxt, nxt + need))
        nxt += need
    g = csr_from_pairs(nxt, np.concatenate(s).astype(np.uint64), np.concatenate(d).astype(np.uint64))
    osym = O.OGraph(g.row_ptr, g.col_idx)
    want = O.diamond(osym)
    with g.to_device(dev) as sym:
        assert SglSolver(sym, "diamond") == want
        assert SglSolver(sym, "diamond", tune=NO_MASKS) == want
        assert SglSolver(sym, "diamond", tune=PER_EDGE) == want
