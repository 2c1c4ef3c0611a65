"""ctypes wrapper of oracle/liboracle.so -- the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


class gmo_graph(C.Structure):
    _fields_ = [("nv", C.c_int32), ("ne", C.c_int64), ("max_degree", C.c_int32),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB)
        G = C.POINTER(gmo_graph)
        P = C.c_void_p
        for name in ("gmo_tc", "gmo_diamond", "gmo_rectangle", "gmo_house", "gmo_pentagon", "gmo_3star", "gmo_4path", "gmo_tailedtriangle",
                     "gmo_alg_bytes_tc", "gmo_alg_bytes_diamond", "gmo_alg_bytes_clique4", "gmo_alg_bytes_motif3"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [G]
        for name in ("gmo_tc_range", "gmo_diamond_range"):
            getattr(L, name).restype = C.c_uint64
            getattr(L, name).argtypes = [G, C.c_int32, C.c_int32]
        L.gmo_tc_sample.restype = C.c_uint64
        L.gmo_tc_sample.argtypes = [G, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
        L.gmo_diamond_sample.restype = C.c_uint64
        L.gmo_diamond_sample.argtypes = [G, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
        L.gmo_clique_sample.restype = C.c_uint64
        L.gmo_clique_sample.argtypes = [G, C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
        L.gmo_motif3_sample.restype = None
        L.gmo_motif3_sample.argtypes = [G, C.c_int32, C.c_int32, P, C.POINTER(C.c_uint64)]
        L.gmo_clique.restype = C.c_uint64
        L.gmo_clique.argtypes = [G, C.c_int]
        L.gmo_clique_range.restype = C.c_uint64
        L.gmo_clique_range.argtypes = [G, C.c_int, C.c_int32, C.c_int32]
        L.gmo_motif3.restype = None
        L.gmo_motif3.argtypes = [G, P]
        L.gmo_motif3_range.restype = None
        L.gmo_motif3_range.argtypes = [G, C.c_int32, C.c_int32, P]
        L.gmo_motif4.restype = None
        L.gmo_motif4.argtypes = [G, P]
        L.gmo_orient.restype = C.c_int
        L.gmo_orient.argtypes = [G, G]
        L.gmo_free.restype = None
        L.gmo_free.argtypes = [G]
        L.gmo_load.restype = C.c_int
        L.gmo_load.argtypes = [C.c_char_p, G]
        L.gmo_edgelist.restype = C.c_int64
        L.gmo_edgelist.argtypes = [G, C.c_int, P, P]
        L.gmo_num_threads.restype = C.c_int
        i32, P32 = C.c_int32, C.c_void_p
        L.gmo_intersect_num.restype = C.c_uint32
        L.gmo_intersect_num.argtypes = [P32, i32, P32, i32]
        L.gmo_intersect_num_upper.restype = C.c_uint32
        L.gmo_intersect_num_upper.argtypes = [P32, i32, P32, i32, i32]
        L.gmo_intersect_set.restype = i32
        L.gmo_intersect_set.argtypes = [P32, i32, P32, i32, P32]
        L.gmo_intersect_set_upper.restype = i32
        L.gmo_intersect_set_upper.argtypes = [P32, i32, P32, i32, i32, P32]
        L.gmo_difference_set.restype = i32
        L.gmo_difference_set.argtypes = [P32, i32, P32, i32, i32, P32]
        L.gmo_difference_set_upper.restype = i32
        L.gmo_difference_set_upper.argtypes = [P32, i32, P32, i32, i32, i32, P32]
        L.gmo_difference_num_upper.restype = C.c_uint32
        L.gmo_difference_num_upper.argtypes = [P32, i32, P32, i32, i32, i32]
        L.gmo_bounded.restype = i32
        L.gmo_bounded.argtypes = [P32, i32, i32]
        _lib = L
    return _lib


class OGraph:
    """numpy-backed graph view handed to the oracle."""

    def __init__(self, row_ptr, col_idx):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        deg = np.diff(self.row_ptr)
        self.c = gmo_graph(self.row_ptr.size - 1, self.col_idx.size, int(deg.max()) if deg.size else 0,
                           self.row_ptr.ctypes.data, self.col_idx.ctypes.data)

    @property
    def nv(self):
        return self.row_ptr.size - 1

    @property
    def ne(self):
        return self.col_idx.size

    def ref(self):
        return C.byref(self.c)


def orient(g: OGraph) -> OGraph:
    out = gmo_graph()
    assert lib().gmo_orient(g.ref(), C.byref(out)) == 0
    nv, ne = out.nv, out.ne
    rp = np.ctypeslib.as_array(C.cast(out.row_ptr, C.POINTER(C.c_int64)), shape=(nv + 1,)).copy()
    ci = np.ctypeslib.as_array(C.cast(out.col_idx, C.POINTER(C.c_int32)), shape=(max(ne, 1),))[:ne].copy()
    lib().gmo_free(C.byref(out))
    return OGraph(rp, ci)


def tc(dag): return int(lib().gmo_tc(dag.ref()))


def tc_sample(dag, stride, offset=0):
    t = C.c_uint64(0)
    c = int(lib().gmo_tc_sample(dag.ref(), stride, offset, C.byref(t)))
    return c, int(t.value)


def diamond_sample(sym, stride, offset=0):
    t = C.c_uint64(0)
    c = int(lib().gmo_diamond_sample(sym.ref(), stride, offset, C.byref(t)))
    return c, int(t.value)


def clique_sample(dag, k, stride, offset=0):
    t = C.c_uint64(0)
    c = int(lib().gmo_clique_sample(dag.ref(), k, stride, offset, C.byref(t)))
    return c, int(t.value)


def motif3_sample(sym, stride, offset=0):
    t = C.c_uint64(0)
    out = (C.c_uint64 * 2)()
    lib().gmo_motif3_sample(sym.ref(), stride, offset, out, C.byref(t))
    return [int(out[0]), int(out[1])], int(t.value)


def alg_bytes(kind, g):
    return int(getattr(lib(), "gmo_alg_bytes_" + kind)(g.ref()))


def num_threads():
    return int(lib().gmo_num_threads())
def diamond(sym): return int(lib().gmo_diamond(sym.ref()))
def rectangle(sym): return int(lib().gmo_rectangle(sym.ref()))
def house(sym): return int(lib().gmo_house(sym.ref()))
def star3(sym): return int(lib().gmo_3star(sym.ref()))
def path4(sym): return int(lib().gmo_4path(sym.ref()))
def tailedtriangle(sym): return int(lib().gmo_tailedtriangle(sym.ref()))
def pentagon(sym): return int(lib().gmo_pentagon(sym.ref()))
def clique(dag, k): return int(lib().gmo_clique(dag.ref(), k))


def motif3(sym):
    out = (C.c_uint64 * 2)()
    lib().gmo_motif3(sym.ref(), out)
    return [int(out[0]), int(out[1])]


def motif4(sym):
    out = (C.c_uint64 * 6)()
    lib().gmo_motif4(sym.ref(), out)
    return [int(x) for x in out]


def ref_binary(name):
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None


def run_ref(name, *args):
    """Run a REAL reference binary (oracle/_ref/<name>) and return its stdout lines."""
    exe = ref_binary(name)
    assert exe, f"{name} not built (oracle/ref/Makefile)"
    env = dict(os.environ, OMP_NUM_THREADS="4")
    out = subprocess.run([exe, *map(str, args)], check=True, capture_output=True, text=True, env=env).stdout
    return out.strip().splitlines()
