"""Python mirror of the reference's four solver entry points over the C ABI.

    void TCSolver    (Graph&, uint64_t& total, int n_gpu, int chunk)            src/triangle/main.cc:5
    void SglSolver   (Graph&, Pattern&, uint64_t& total, int n_dev, int chunk)  src/sgl/main.cc:7
    void CliqueSolver(Graph&, int k, uint64_t& total, int, int)                 src/clique/main.cc:6
    void MotifSolver (Graph&, int k, std::vector<uint64_t>&, int, int)          src/motif/main.cc:7

Same names, argument meaning and error behaviour; results are returned instead of written
through references. Each call goes straight to the HIP library -- no host compute path.
Multi-GPU: one process per GPU; pass ``rank``/``world`` (the task-chunk share) and sum the
returned per-rank counts with one all-reduce (see graphminer_amd.dist).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from . import _lib
from ._lib import GM_ERR_UNSUPPORTED, gm_launch, gm_stats
from .graph import DeviceGraph

# include/pattern.hh:4-15
num_possible_patterns = [0, 1, 1, 2, 6, 21, 112, 853, 11117, 261080]


@dataclass
class Stats:
    kernel_ms: float = 0.0
    tasks: int = 0
    chunks: int = 0
    grid: int = 0
    block: int = 0


def _launch(rank=0, world=1, chunk=0, policy=_lib.GM_PART_ROUND_ROBIN, stream=0, d_counts=0, tune=None):
    la = gm_launch()
    la.stream = stream or None
    la.rank, la.world, la.policy, la.chunk = rank, world, policy, chunk
    la.d_counts = d_counts or None
    for i, t in enumerate(tune or []):
        la.tune[i] = int(t)
    return la


def _stats(st: gm_stats) -> Stats:
    return Stats(st.kernel_ms, int(st.tasks), int(st.chunks), int(st.grid), int(st.block))


def TCSolver(g: DeviceGraph, *, rank=0, world=1, chunk=0, return_stats=False, **kw):
    """Triangle count of an ORIENTED graph: sum over DAG edges (u,v) of |N+(u) ^ N+(v)|."""
    lib = _lib.load()
    la, st, total = _launch(rank, world, chunk, **kw), gm_stats(), C.c_uint64(0)
    _lib.check(lib.gm_tc(g.handle, C.byref(la), C.byref(total), C.byref(st)), "gm_tc")
    return (int(total.value), _stats(st)) if return_stats else int(total.value)


def tc_core_info(g: DeviceGraph) -> dict:
    """the hub corner the triangle count of this ORIENTED graph takes on the matrix cores (gm_tc_core_info; after a first TCSolver call)"""
    info = (C.c_int64 * 4)()
    _lib.check(_lib.load().gm_tc_core_info(g.handle, info), "gm_tc_core_info")
    return {"h": int(info[0]), "edges": int(info[1]), "blocks": int(info[2]), "core_h": int(info[3])}


def SglSolver(g: DeviceGraph, pattern: str, *, rank=0, world=1, chunk=0, return_stats=False, **kw):
    """Edge-induced subgraph listing on the SYMMETRIC graph; pattern by name (include/pattern.hh:62-78).

    Unknown / unimplemented names behave like the reference's omp solver: "Not implemented",
    total 0 (src/sgl/omp_base.cc:51-53)."""
    lib = _lib.load()
    la, st, total = _launch(rank, world, chunk, **kw), gm_stats(), C.c_uint64(0)
    rc = lib.gm_sgl(g.handle, pattern.encode(), C.byref(la), C.byref(total), C.byref(st))
    if rc == GM_ERR_UNSUPPORTED:
        print("Not implemented")
        return (0, Stats()) if return_stats else 0
    _lib.check(rc, "gm_sgl")
    return (int(total.value), _stats(st)) if return_stats else int(total.value)


def CliqueSolver(g: DeviceGraph, k: int, *, rank=0, world=1, chunk=0, return_stats=False, **kw):
    """k-clique count on the ORIENTED graph."""
    lib = _lib.load()
    la, st, total = _launch(rank, world, chunk, **kw), gm_stats(), C.c_uint64(0)
    _lib.check(lib.gm_clique(g.handle, k, C.byref(la), C.byref(total), C.byref(st)), "gm_clique")
    return (int(total.value), _stats(st)) if return_stats else int(total.value)


def MotifSolver(g: DeviceGraph, k: int, *, rank=0, world=1, chunk=0, return_stats=False, formula=False, **kw):
    """k-motif counts on the SYMMETRIC graph; k=3 -> [wedges, triangles] (CPU order).

    formula=True is the reference's motif_omp_formula / motif_gpu_formula variant (enumerate triangles only,
    derive the wedges); with world > 1 its per-rank wedge value is a partial modulo 2**64 -- sum the ranks."""
    lib = _lib.load()
    n = num_possible_patterns[k] if 0 <= k < len(num_possible_patterns) else 0
    la, st = _launch(rank, world, chunk, **kw), gm_stats()
    out = (C.c_uint64 * max(n, 1))()
    fn = lib.gm_motif_formula if formula else lib.gm_motif
    _lib.check(fn(g.handle, k, C.byref(la), out, n, C.byref(st)), "gm_motif")
    res = [int(out[i]) for i in range(n)]
    return (res, _stats(st)) if return_stats else res


# ---- tailedtriangle / 4path / 3star on several ranks (include/graphminer_amd.h: gm_sgl4_*) ------------------------------------------------
def sgl4_partial(g: DeviceGraph, *, rank=0, world=1, chunk=0, return_stats=False, **kw):
    """this rank's share of the four per-edge sums the three patterns are closed forms of; sum the ranks' lists, then sgl4_finish"""
    la, st, raw = _launch(rank, world, chunk, **kw), gm_stats(), (C.c_uint64 * 4)()
    _lib.check(_lib.load().gm_sgl4_partial(g.handle, C.byref(la), None if la.d_counts else raw, C.byref(st)), "gm_sgl4_partial")
    res = [int(x) for x in raw]
    return (res, _stats(st)) if return_stats else res


def sgl4_finish(pattern: str, raw) -> int:
    """summed (mod 2**64) raw sums of every rank -> the count of `pattern`"""
    total = C.c_uint64(0)
    _lib.check(_lib.load().gm_sgl4_finish(pattern.encode(), (C.c_uint64 * 4)(*[int(x) & (2**64 - 1) for x in raw]), C.byref(total)), "gm_sgl4_finish")
    return int(total.value)


# ---- diamond on several ranks with the one-GPU algorithm (include/graphminer_amd.h: gm_diamond_support_*) ------------------------------
def diamond_support_size(g: DeviceGraph, world: int = 1) -> int:
    """uint32 entries of a rank's support array: |E+| of the oriented copy, padded so that every rank's reduce-scatter slice is equal"""
    n = C.c_int64(0)
    _lib.check(_lib.load().gm_diamond_support_size(g.handle, world, C.byref(n)), "gm_diamond_support_size")
    return int(n.value)


def diamond_support_partial(g: DeviceGraph, d_support: int, n_entries: int, *, rank=0, world=1, chunk=0, return_stats=False, **kw):
    """this rank's share of the triangle pass adds its increments into the caller's DEVICE buffer (zeroed by the call); asynchronous on
    `stream` when `d_counts` is given"""
    la, st = _launch(rank, world, chunk, **kw), gm_stats()
    _lib.check(_lib.load().gm_diamond_support_partial(g.handle, C.byref(la), d_support, n_entries, C.byref(st)), "gm_diamond_support_partial")
    return _stats(st) if return_stats else None


def diamond_support_finish(g: DeviceGraph, d_support: int, count: int, *, return_stats=False, **kw):
    """sum C(t, 2) over `count` reduced support entries at `d_support` (a rank's slice) -> this rank's part of the diamond count"""
    la, st, total = _launch(0, 1, 0, **kw), gm_stats(), C.c_uint64(0)
    _lib.check(_lib.load().gm_diamond_support_finish(g.handle, C.byref(la), d_support, count, None if la.d_counts else C.byref(total), C.byref(st)),
               "gm_diamond_support_finish")
    return (int(total.value), _stats(st)) if return_stats else int(total.value)
