"""ctypes binding of include/graphminer_amd.h (the C-ABI shared library).

There is no fallback: if ``libgraphminer_amd.so`` has not been built (``python -c
"import __graft_entry__ as g; g.build()"`` or ``make -C graphminer_amd``) importing the
compute API raises ``GraphMinerBuildError``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "libgraphminer_amd.so")  # GM_LIB_PATH: A/B builds


class GraphMinerBuildError(RuntimeError):
    pass


class GraphMinerError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: gm_status {status} ({detail})")


class gm_csr(C.Structure):
    _fields_ = [
        ("nv", C.c_int32),
        ("ne", C.c_int64),
        ("max_deg", C.c_int32),
        ("row_ptr", C.c_void_p),
        ("col_idx", C.c_void_p),
    ]


class gm_launch(C.Structure):
    _fields_ = [
        ("stream", C.c_void_p),
        ("rank", C.c_int32),
        ("world", C.c_int32),
        ("policy", C.c_int32),
        ("chunk", C.c_int32),
        ("d_counts", C.c_void_p),
        ("tune", C.c_int32 * 8),
    ]


class gm_stats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double),
        ("tasks", C.c_uint64),
        ("chunks", C.c_uint64),
        ("grid", C.c_uint32),
        ("block", C.c_uint32),
    ]


class gm_setup_times(C.Structure):
    _fields_ = [("orient_ms", C.c_double), ("table_ms", C.c_double), ("bitmap_ms", C.c_double), ("relabel_ms", C.c_double),
                ("other_ms", C.c_double)]


GM_OK, GM_ERR_INVALID, GM_ERR_NO_DEVICE, GM_ERR_HIP, GM_ERR_TOO_LARGE, GM_ERR_UNSUPPORTED, GM_ERR_IO, GM_ERR_FORMAT = range(8)
GM_PART_ROUND_ROBIN, GM_PART_RANGE, GM_PART_VERTEX = 0, 1, 2
(GM_OP_INTERSECT_NUM, GM_OP_INTERSECT_NUM_UPPER, GM_OP_INTERSECT_SET, GM_OP_DIFFERENCE_NUM,
 GM_OP_DIFFERENCE_NUM_UPPER, GM_OP_DIFFERENCE_SET, GM_OP_INTERSECT_SET_UPPER, GM_OP_DIFFERENCE_SET_UPPER, GM_OP_COUNT_SMALLER) = range(9)

# every symbol include/graphminer_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("gm_strerror", C.c_char_p, [C.c_int]),
    ("gm_last_error", C.c_char_p, []),
    ("gm_version", C.c_int, []),
    ("gm_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("gm_graph_upload", C.c_int, [C.POINTER(gm_csr), C.c_int, C.POINTER(_P)]),
    ("gm_graph_from_device", C.c_int, [C.c_int32, C.c_int64, _P, _P, C.c_int, C.POINTER(_P)]),
    ("gm_graph_orient", C.c_int, [_P, C.POINTER(_P)]),
    ("gm_graph_sort_neighbors", C.c_int, [_P]),
    ("gm_graph_meta", C.c_int, [_P, C.POINTER(gm_csr)]),
    ("gm_graph_download", C.c_int, [_P, _P, _P]),
    ("gm_graph_renumbered", C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    ("gm_graph_free", None, [_P]),
    ("gm_partition", C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                               C.POINTER(C.c_int64)]),
    ("gm_chunk_table", C.c_int, [C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    ("gm_graph_setup_times", C.c_int, [_P, C.POINTER(gm_setup_times)]),
    ("gm_kernel_times", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("gm_corner_times", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("gm_tc", C.c_int, [_P, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_sgl", C.c_int, [_P, C.c_char_p, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_clique", C.c_int, [_P, C.c_int, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_motif", C.c_int, [_P, C.c_int, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.c_int, C.POINTER(gm_stats)]),
    ("gm_motif4_partial", C.c_int, [_P, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_motif4_finish", C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("gm_sgl4_partial", C.c_int, [_P, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_sgl4_finish", C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("gm_motif_formula", C.c_int, [_P, C.c_int, C.POINTER(gm_launch), C.POINTER(C.c_uint64), C.c_int, C.POINTER(gm_stats)]),
    ("gm_setop_batch", C.c_int, [C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("gm_rmat_keys", C.c_int, [C.c_int, C.c_int64, C.c_uint64, _P, _P]),
    ("gm_clique4_level2_bytes", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("gm_clique4_gather_info", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("gm_calib_stream", C.c_int, [_P, C.c_int64, _P, _P]),
    ("gm_stream_ceiling", C.c_int, [_P, C.c_int64, _P, _P]),
    ("gm_issue_calib", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("gm_diamond_support_size", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("gm_diamond_support_info", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("gm_tc_core_info", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("gm_sup_core_info", C.c_int, [_P, C.POINTER(C.c_int64)]),
    ("gm_diamond_support_partial", C.c_int, [_P, C.POINTER(gm_launch), _P, C.c_int64, C.POINTER(gm_stats)]),
    ("gm_diamond_support_finish", C.c_int, [_P, C.POINTER(gm_launch), _P, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(gm_stats)]),
    ("gm_constant", C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    ("gm_selftest", C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    ("gm_dev_option", C.c_int, [C.c_char_p, C.c_char_p]),
    ("gm_dev_option_get", C.c_char_p, [C.c_char_p]),
]

_lib = None


def load():
    """Load the shared library (once) and declare all prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GraphMinerBuildError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C graphminer_amd`). "
            "graphminer_amd has no CPU fallback."
        )
    try:
        # When torch is (or will be) in the process, its bundled libamdhip64 must be the one
        # that resolves our HIP symbols, so import it first if it is importable.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing only
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dev_option(name, value=None):
    """Set (value) / remove (None) a developer option of the library; name = None removes all (include/graphminer_amd.h gm_dev_option:
    the switches tests and A/B runs use -- the library reads no algorithm switch from the environment)."""
    lib = load()
    st = lib.gm_dev_option(None if name is None else str(name).encode(), None if value is None else str(value).encode())
    if st != GM_OK:
        raise GraphMinerError(st, "gm_dev_option", str(name))


def check(status: int, where: str):
    if status != GM_OK:
        lib = load()
        detail = lib.gm_strerror(status).decode()
        if status in (GM_ERR_HIP, GM_ERR_NO_DEVICE, GM_ERR_TOO_LARGE):
            last = lib.gm_last_error().decode()
            if last:
                detail += "; " + last
        raise GraphMinerError(status, where, detail)
