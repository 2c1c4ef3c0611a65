"""Host-side graph container and the device handle.

``Graph`` mirrors the accessors of the reference's ``Graph`` class
(include/graph.h:49-148, loader src/common/graph.cc:4-42): it reads the three-file
format ``<prefix>.meta.txt`` / ``<prefix>.vertex.bin`` (int64 row_ptr[nv+1]) /
``<prefix>.edge.bin`` (int32 col_idx[ne]) and enforces the same asserts
(graph.cc:30-34). It only HOLDS data; every computation (orientation, mining)
happens on the GPU through ``DeviceGraph`` -- there is no host compute path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import GraphMinerError, gm_csr


class GraphFormatError(ValueError):
    pass


class Graph:
    """CSR graph in host memory (vertex ids int32, offsets int64; include/common.h:36-37)."""

    def __init__(self, prefix: str | None = None, *, row_ptr=None, col_idx=None, name: str = ""):
        self.meta = {}
        if prefix is not None:
            self._load(prefix)
        else:
            rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
            ci = np.ascontiguousarray(col_idx, dtype=np.int32)
            if rp.ndim != 1 or rp.size < 1 or rp[0] != 0 or rp[-1] != ci.size:
                raise GraphFormatError("row_ptr must start at 0 and end at len(col_idx)")
            self.row_ptr, self.col_idx = rp, ci
            self.name = name
            self.path = ""
        deg = np.diff(self.row_ptr)
        if deg.size and deg.min() < 0:
            raise GraphFormatError("row_ptr is not monotone")
        self.max_degree = int(deg.max()) if deg.size else 0

    # --- loader (src/common/graph.cc:13-42) -------------------------------------------------
    def _load(self, prefix: str):
        i = prefix.rfind("/")
        self.path = prefix[:i] if i >= 0 else ""
        j = self.path.rfind("/")
        self.name = self.path[j + 1:] if j >= 0 else ""
        meta_path = prefix + ".meta.txt"
        if not os.path.exists(meta_path):
            raise FileNotFoundError(meta_path)  # custom_alloc.h:38-41: exit(1) in the reference
        with open(meta_path) as f:
            tok = f.read().split()
        if len(tok) < 7:
            raise GraphFormatError("meta.txt: expected nv ne vid_size eid_size vlabel_size elabel_size max_degree ...")
        nv, ne, vid_size, eid_size = int(tok[0]), int(tok[1]), int(tok[2]), int(tok[3])
        max_degree = int(tok[6])
        if vid_size != 4:
            raise GraphFormatError("sizeof(vidType) must be 4 (graph.cc:30)")
        if eid_size != 8:
            raise GraphFormatError("sizeof(eidType) must be 8 (graph.cc:31)")
        if not (0 < max_degree < nv):
            raise GraphFormatError("need 0 < max_degree < nv (graph.cc:34)")
        self.meta = {
            "feat_len": int(tok[7]) if len(tok) > 7 else 0,
            "num_vertex_classes": int(tok[8]) if len(tok) > 8 else 0,
            "num_edge_classes": int(tok[9]) if len(tok) > 9 else 0,
        }
        rp = np.fromfile(prefix + ".vertex.bin", dtype=np.int64)
        ci = np.fromfile(prefix + ".edge.bin", dtype=np.int32)
        if rp.size != nv + 1 or ci.size != ne:
            raise GraphFormatError("vertex.bin / edge.bin size does not match meta.txt")
        if rp[0] != 0 or rp[-1] != ne:
            raise GraphFormatError("row_ptr must start at 0 and end at ne")
        self.row_ptr, self.col_idx = rp, ci

    def save(self, prefix: str):
        """Write the three-file format (the inverse of the loader)."""
        os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
        with open(prefix + ".meta.txt", "w") as f:
            f.write(f"{self.V()}\n{self.E()}\n4 8 1 2\n{self.max_degree}\n0\n0\n0\n")
        self.row_ptr.astype(np.int64).tofile(prefix + ".vertex.bin")
        self.col_idx.astype(np.int32).tofile(prefix + ".edge.bin")

    # --- accessors (include/graph.h:63-84) --------------------------------------------------
    def V(self) -> int:
        return int(self.row_ptr.size - 1)

    def E(self) -> int:
        return int(self.col_idx.size)

    def get_max_degree(self) -> int:
        return self.max_degree

    def get_degree(self, v: int) -> int:
        return int(self.row_ptr[v + 1] - self.row_ptr[v])

    def N(self, v: int) -> np.ndarray:
        return self.col_idx[self.row_ptr[v]:self.row_ptr[v + 1]]

    def out_rowptr(self) -> np.ndarray:
        return self.row_ptr

    def out_colidx(self) -> np.ndarray:
        return self.col_idx

    def print_meta_data(self) -> str:
        # Graph::print_meta_data, src/common/graph.cc:645-647
        return f"|V|: {self.V()}, |E|: {self.E()}, Max Degree: {self.max_degree}"

    def to_device(self, device: int = 0) -> "DeviceGraph":
        return DeviceGraph.upload(self, device)


class DeviceGraph:
    """Opaque gm_graph handle (replaces GraphGPU, include/graph_gpu.h:6-211)."""

    def __init__(self, handle: int, device: int, keepalive=None):
        self._h = C.c_void_p(handle)
        self.device = device
        self._keep = keepalive  # borrowed device tensors must outlive the handle
        m = gm_csr()
        _lib.check(_lib.load().gm_graph_meta(self._h, C.byref(m)), "gm_graph_meta")
        self.nv, self.ne, self.max_degree = int(m.nv), int(m.ne), int(m.max_deg)

    @classmethod
    def upload(cls, g: Graph, device: int = 0) -> "DeviceGraph":
        lib = _lib.load()
        csr = gm_csr(g.V(), g.E(), g.max_degree, g.row_ptr.ctypes.data, g.col_idx.ctypes.data)
        h = C.c_void_p()
        _lib.check(lib.gm_graph_upload(C.byref(csr), device, C.byref(h)), "gm_graph_upload")
        return cls(h.value, device)

    @classmethod
    def from_device_ptrs(cls, nv: int, ne: int, d_row_ptr: int, d_col_idx: int, device: int = 0, keepalive=None):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.gm_graph_from_device(nv, ne, d_row_ptr, d_col_idx, device, C.byref(h)), "gm_graph_from_device")
        return cls(h.value, device, keepalive)

    def orient(self) -> "DeviceGraph":
        """Graph::orientation (src/common/graph.cc:233-279) on the GPU."""
        h = C.c_void_p()
        _lib.check(_lib.load().gm_graph_orient(self._h, C.byref(h)), "gm_graph_orient")
        return DeviceGraph(h.value, self.device)

    def sort_neighbors(self) -> "DeviceGraph":
        """Graph::sort_neighbors (src/common/graph.cc:138-146) on the GPU, in place; returns self."""
        _lib.check(_lib.load().gm_graph_sort_neighbors(self._h), "gm_graph_sort_neighbors")
        return self

    def renumbered(self, mode: int) -> Graph:
        """the renumbered copy the solvers take (gm_graph_renumbered: 0 / 1 by degree, 2 topological), downloaded; the device copy
        stays with this handle"""
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.gm_graph_renumbered(self._h, mode, C.byref(h)), "gm_graph_renumbered")
        m = gm_csr()
        _lib.check(lib.gm_graph_meta(h, C.byref(m)), "gm_graph_meta")
        rp = np.empty(int(m.nv) + 1, dtype=np.int64)
        ci = np.empty(int(m.ne), dtype=np.int32)
        _lib.check(lib.gm_graph_download(h, rp.ctypes.data, ci.ctypes.data), "gm_graph_download")
        return Graph(row_ptr=rp, col_idx=ci)

    def download(self) -> Graph:
        rp = np.empty(self.nv + 1, dtype=np.int64)
        ci = np.empty(self.ne, dtype=np.int32)
        _lib.check(_lib.load().gm_graph_download(self._h, rp.ctypes.data, ci.ctypes.data), "gm_graph_download")
        return Graph(row_ptr=rp, col_idx=ci)

    def V(self) -> int:
        return self.nv

    def E(self) -> int:
        return self.ne

    def get_max_degree(self) -> int:
        return self.max_degree

    def kernel_times_ms(self, n: int = 64):
        """HIP-event durations of the last n mining-kernel launches (caller synchronised the stream)."""
        buf = (C.c_double * max(n, 1))()
        got = C.c_int(0)
        _lib.check(_lib.load().gm_kernel_times(self.handle, n, buf, C.byref(got)), "gm_kernel_times")
        return [float(buf[i]) for i in range(got.value)]

    def corner_times_ms(self, n: int = 64):
        """the part of kernel_times_ms() spent in the launches' hub-corner kernel on the matrix cores (gm_corner_times; 0: none)"""
        buf = (C.c_double * max(n, 1))()
        got = C.c_int(0)
        _lib.check(_lib.load().gm_corner_times(self.handle, n, buf, C.byref(got)), "gm_corner_times")
        return [float(buf[i]) for i in range(got.value)]

    def setup_times_ms(self) -> dict:
        """Accumulated pre-processing time of this handle (gm_graph_setup_times): orientation, task tables, hub
        bitmaps, renumbered copies, per-pattern tables -- the steps the reference leaves untimed."""
        from ._lib import gm_setup_times

        t = gm_setup_times()
        _lib.check(_lib.load().gm_graph_setup_times(self.handle, C.byref(t)), "gm_graph_setup_times")
        return {k: round(float(getattr(t, k)), 3) for k, _ in gm_setup_times._fields_}

    def free(self):
        if self._h:
            _lib.load().gm_graph_free(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()
        return False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise GraphMinerError(1, "DeviceGraph", "handle already freed")
        return self._h
