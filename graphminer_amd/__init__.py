"""graphminer_amd -- MI355X-native subgraph-matching hot path (TC / diamond / k-clique / 3-motif).

The product is the C-ABI shared library ``libgraphminer_amd.so`` (include/graphminer_amd.h,
hand-written HIP for gfx950). This package is the thin host-side mirror of the reference's
solver interface over that library; it contains no compute of its own and no CPU fallback.
"""
from ._lib import (GM_PART_RANGE, GM_PART_ROUND_ROBIN, GraphMinerBuildError, GraphMinerError, LIB_PATH, load)
from .graph import DeviceGraph, Graph, GraphFormatError
from .solvers import CliqueSolver, MotifSolver, SglSolver, Stats, TCSolver, num_possible_patterns

__all__ = [
    "Graph", "DeviceGraph", "GraphFormatError", "GraphMinerError", "GraphMinerBuildError",
    "TCSolver", "SglSolver", "CliqueSolver", "MotifSolver", "Stats", "num_possible_patterns",
    "GM_PART_ROUND_ROBIN", "GM_PART_RANGE", "LIB_PATH", "load",
]
