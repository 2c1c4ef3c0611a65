"""Deterministic R-MAT graph generator (tooling).

The reference ships no generator (its converter lives in another repo, README.md:104), so this
module DEFINES the synthetic fixtures: edge i of ``edge_factor * 2**scale`` draws one quadrant
per bit level with probabilities (a,b,c,d) = (0.57, 0.19, 0.19, 0.05) from a counter-based
SplitMix64 hash of (seed, i, level); then self-loops are dropped, the edge set is symmetrised,
rows are sorted and de-duplicated, max_degree = longest row.

Two implementations of the SAME stream:
  * ``rmat_csr_numpy``  -- host, vectorised numpy (tests, small scales);
  * ``rmat_csr_device`` -- the HIP kernel ``gm_rmat_keys`` + torch sort/unique (bench, full scale).
tests/test_gpu_parity.py::test_rmat_device_generator_equals_numpy checks they produce identical CSR arrays.
"""
from __future__ import annotations

import numpy as np

from .graph import DeviceGraph, Graph

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G1 = np.uint64(0x9E3779B97F4A7C15)
_G2 = np.uint64(0xD1B54A32D192ED03)
TA = 2448131358  # floor(0.57 * 2^32)
TB = 3264175144  # TA + floor(0.19 * 2^32)
TC = 4080218930  # TB + floor(0.19 * 2^32)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def rmat_edges_numpy(scale: int, edge_factor: int, seed: int = 42):
    """(src, dst) uint64 arrays of the raw generated pairs (self-loops included)."""
    n = (1 << scale) * edge_factor
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        h = _mix64(np.uint64(seed) + _G1 * i)
        s = np.zeros(n, dtype=np.uint64)
        d = np.zeros(n, dtype=np.uint64)
        for l in range(scale):
            r = _mix64(h + _G2 * np.uint64(l + 1)) >> np.uint64(32)
            q = (r >= TA).astype(np.uint64) + (r >= TB).astype(np.uint64) + (r >= TC).astype(np.uint64)
            s = (s << np.uint64(1)) | (q >> np.uint64(1))
            d = (d << np.uint64(1)) | (q & np.uint64(1))
    return s, d


def csr_from_pairs(nv: int, s: np.ndarray, d: np.ndarray) -> Graph:
    """Drop self-loops, symmetrise, sort + dedupe rows."""
    keep = s != d
    s, d = s[keep].astype(np.uint64), d[keep].astype(np.uint64)
    keys = np.concatenate([(s << np.uint64(32)) | d, (d << np.uint64(32)) | s])
    keys = np.unique(keys)
    src = (keys >> np.uint64(32)).astype(np.int64)
    col = (keys & np.uint64(0xFFFFFFFF)).astype(np.int32)
    row_ptr = np.zeros(nv + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=nv), out=row_ptr[1:])
    return Graph(row_ptr=row_ptr, col_idx=col, name=f"rmat")


def rmat_csr_numpy(scale: int, edge_factor: int, seed: int = 42) -> Graph:
    s, d = rmat_edges_numpy(scale, edge_factor, seed)
    g = csr_from_pairs(1 << scale, s, d)
    g.name = f"rmat{scale}_ef{edge_factor}_s{seed}"
    return g


def rmat_csr_device(scale: int, edge_factor: int, seed: int = 42, device: int = 0):
    """Generate on the GPU. Returns (DeviceGraph, row_ptr_tensor, col_idx_tensor); the tensors
    own the memory the handle borrows (kept alive by the handle too)."""
    import ctypes as C

    import torch

    from . import _lib

    lib = _lib.load()
    nv = 1 << scale
    n = nv * edge_factor
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        keys = torch.empty(2 * n, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.gm_rmat_keys(scale, n, C.c_uint64(seed), keys.data_ptr(), stream), "gm_rmat_keys")
        # sort as UNSIGNED 64-bit: sentinel 0xFFFF.. (self-loop) is -1 as int64 and sorts first; keys are < 2^62
        if keys.numel() < (1 << 31):
            keys = torch.unique(keys)  # sorted ascending (signed): [-1?, k0, k1, ...]
            if keys.numel() and int(keys[0].item()) == -1:
                keys = keys[1:]
        else:
            # more keys than the device sort takes in one call (2^31 items): sort / dedupe the eight ranges of the three leading
            # source bits one after the other (R-MAT puts at most 0.76^3 = 44 % of the keys into one of them) and concatenate
            sh = 32 + max(scale - 3, 0)
            step = 1 << 30  # (boolean indexing of more than 2^31 elements overflows inside torch as well: slice first)
            buckets = [[] for _ in range(8)]
            for lo in range(0, keys.numel(), step):
                sl = keys[lo:lo + step]
                top = sl >> sh
                for b in range(8):
                    buckets[b].append(sl[(sl >= 0) & (top == b)])
                del sl, top
            del keys
            parts = []
            for b in range(8):
                parts.append(torch.unique(torch.cat(buckets[b])))
                buckets[b] = None
            keys = torch.cat(parts)
            del parts, buckets
        src = keys >> 32
        col = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
        deg = torch.bincount(src, minlength=nv)
        row_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        torch.cumsum(deg, 0, out=row_ptr[1:])
        del keys, src, deg
        torch.cuda.synchronize(dev)
    g = DeviceGraph.from_device_ptrs(nv, int(col.numel()), row_ptr.data_ptr(), col.data_ptr(), device,
                                     keepalive=(row_ptr, col))
    return g, row_ptr, col


def uniform_csr_device(nv: int, m: int, seed: int = 1, device: int = 0):
    """Erdos-Renyi-like graph on the GPU (torch RNG): m random pairs, self-loops dropped, symmetrised, deduplicated.
    A second synthetic stand-in with LiveJournal's SIZE but a flat degree profile (mean oriented list ~9): the
    short-list regime, where R-MAT (hub-dominated) is the long-list regime. Not bit-reproducible across torch
    versions -- used for measurements only, never for goldens."""
    import torch

    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        s = torch.randint(0, nv, (m,), device=dev, generator=gen)
        d = torch.randint(0, nv, (m,), device=dev, generator=gen)
        keep = s != d
        s, d = s[keep], d[keep]
        keys = torch.unique(torch.cat([(s << 32) | d, (d << 32) | s]))
        src = keys >> 32
        col = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
        row_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(src, minlength=nv), 0, out=row_ptr[1:])
        del keys, src, s, d
        torch.cuda.synchronize(dev)
    g = DeviceGraph.from_device_ptrs(nv, int(col.numel()), row_ptr.data_ptr(), col.data_ptr(), device, keepalive=(row_ptr, col))
    return g, row_ptr, col


def powerlaw_csr_device(nv: int, m: int, max_deg: int = 20000, gamma: float = 2.5, seed: int = 1, device: int = 0):
    """Chung-Lu power-law graph on the GPU (torch RNG): m endpoint pairs drawn from weights w_i ~ (i + i0)^(-1/(gamma-1)),
    i0 chosen so that the heaviest vertex expects `max_deg` neighbours; ids randomly permuted (hubs are not clustered at low
    ids, unlike R-MAT); self-loops dropped, symmetrised, de-duplicated. With nv = 4,847,571, m = 43,000,000,
    max_deg = 20,000 it has LiveJournal's published |V|, |E| and maximum degree (src/triangle/README.md:58): the third
    stand-in for BASELINE configs 2 / 3, between the flat uniform graph and the hub-dominated R-MAT. Measurements only
    (not bit-reproducible across torch versions), never goldens."""
    import torch

    alpha = 1.0 / (gamma - 1.0)
    one = 1.0 - alpha

    def d0(i0):  # expected degree of the heaviest vertex
        w = ((nv + i0) ** one - i0 ** one) / one
        return 2.0 * m * (i0 ** -alpha) / w

    lo, hi = 1e-6, float(nv)
    for _ in range(200):  # d0 decreases in i0
        mid = (lo * hi) ** 0.5
        lo, hi = (mid, hi) if d0(mid) > max_deg else (lo, mid)
    i0 = hi
    dev = torch.device("cuda", device)
    with torch.cuda.device(dev):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        a_, b_ = i0 ** one, (nv + i0) ** one

        def draw():
            u = torch.rand(m, device=dev, dtype=torch.float64, generator=gen)
            x = (u * (b_ - a_) + a_) ** (1.0 / one) - i0
            return x.floor().clamp_(0, nv - 1).to(torch.int64)

        perm = torch.randperm(nv, device=dev, generator=gen)
        s, d = perm[draw()], perm[draw()]
        keep = s != d
        s, d = s[keep], d[keep]
        keys = torch.unique(torch.cat([(s << 32) | d, (d << 32) | s]))
        src = keys >> 32
        col = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
        row_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(src, minlength=nv), 0, out=row_ptr[1:])
        del keys, src, s, d, perm
        torch.cuda.synchronize(dev)
    g = DeviceGraph.from_device_ptrs(nv, int(col.numel()), row_ptr.data_ptr(), col.data_ptr(), device, keepalive=(row_ptr, col))
    return g, row_ptr, col


def community_csr_device(nv: int, m: int, csize: int = 64, p_in: float = 0.6, max_deg: int = 20000, seed: int = 1, device: int = 0):
    """A stand-in with LiveJournal's TRIANGLE DENSITY: the flat and the power-law stand-ins have LiveJournal's |V| and |E|
    but 4 M / 30 M triangles where LiveJournal has 286 M (src/triangle/README.md:58: 6.7 per oriented edge).  Planted communities -- cliques
    of `csize` vertices thinned to density `p_in` (64, 0.6: ~1210 edges and ~9000 triangles each) -- take as many vertices as 90 % of the m
    edges need; the other 10 % are a Chung-Lu power-law background over ALL vertices (hubs up to max_deg scaled by that share), ids randomly
    permuted.  nv = 4,847,571, m = 43,000,000 gives ~2.0 M community vertices and ~290 M triangles.  Measurements only (torch RNG)."""
    import torch

    dev = torch.device("cuda", device)
    per_c = csize * (csize - 1) // 2
    m_in = int(m * 0.9)
    ncomm = min(int(m_in / (per_c * p_in)), nv // csize)
    with torch.cuda.device(dev):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        iu = torch.triu_indices(csize, csize, 1, device=dev)
        base = (torch.arange(ncomm, device=dev, dtype=torch.int64) * csize).repeat_interleave(per_c)
        a_ = iu[0].repeat(ncomm) + base
        b_ = iu[1].repeat(ncomm) + base
        keep = torch.rand(a_.numel(), device=dev, generator=gen) < p_in
        a_, b_ = a_[keep], b_[keep]
        del base, keep
        perm = torch.randperm(nv, device=dev, generator=gen)
        s1, d1 = perm[a_], perm[b_]
        del a_, b_
    # background: the power-law generator's pairs over all vertices
    m_bg = max(m - int(s1.numel()), 0)
    gbg, rp_bg, ci_bg = powerlaw_csr_device(nv, max(m_bg, 1), max(int(max_deg), 2), 2.5, seed + 1, device)
    with torch.cuda.device(dev):
        src_bg = torch.repeat_interleave(torch.arange(nv, device=dev, dtype=torch.int64), (rp_bg[1:] - rp_bg[:-1]))
        k_bg = (src_bg << 32) | ci_bg.to(torch.int64)
        gbg.free()
        del rp_bg, ci_bg, src_bg
        keys = torch.unique(torch.cat([(s1 << 32) | d1, (d1 << 32) | s1, k_bg]))
        del s1, d1, k_bg
        src = keys >> 32
        col = (keys & 0xFFFFFFFF).to(torch.int32).contiguous()
        row_ptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(src, minlength=nv), 0, out=row_ptr[1:])
        del keys, src
        torch.cuda.synchronize(dev)
    g = DeviceGraph.from_device_ptrs(nv, int(col.numel()), row_ptr.data_ptr(), col.data_ptr(), device, keepalive=(row_ptr, col))
    return g, row_ptr, col
