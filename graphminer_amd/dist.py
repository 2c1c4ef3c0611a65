"""Multi-GPU glue: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests) for the ONE collective the path has -- the all-reduce of the 64-bit
match counts (replaces the host-side sum of src/clique/multigpu.cu:134 and
MPI_Allreduce of src/triangle/dist_cpu.cpp:56). The CSR is replicated; tasks are split by chunk id.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


_M64 = (1 << 64) - 1


def allreduce_counts(counts, device=None):
    """Sum per-rank uint64 counts over all ranks, modulo 2**64. `counts`: list of ints (returned as a list of
    ints in [0, 2**64)) or an int64 tensor (reduced in place and returned as given).

    Per-rank partials are only defined modulo 2**64 and routinely exceed 2**63: the 3-motif wedge partial
    (c[2] - c[0]), gm_motif_formula on ranks != 0 (0 - 3T) and the signed pentagon partial. torch has no uint64
    all-reduce, so the values travel as their two's-complement int64 image -- int64 addition wraps exactly like
    uint64 addition -- and are mapped back to [0, 2**64) afterwards."""
    import torch
    import torch.distributed as dist

    if isinstance(counts, torch.Tensor):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(counts)
        return counts
    vals = [int(c) & _M64 for c in counts]
    t = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v for v in vals], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return [int(x) & _M64 for x in t.cpu().tolist()]


def partition(n_chunks: int, rank: int, world: int, policy: int = _lib.GM_PART_ROUND_ROBIN):
    """Chunk ids owned by `rank`: the library's own index arithmetic (gm_partition)."""
    f, s, c = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(_lib.load().gm_partition(n_chunks, rank, world, policy, C.byref(f), C.byref(s), C.byref(c)), "gm_partition")
    return range(f.value, f.value + s.value * c.value, s.value)


def chunk_table(row_ptr, chunk: int = 0, for_clique: bool = False) -> np.ndarray:
    """The solvers' task-chunk table for a CSR as an (n,4) int32 array {u_begin,u_end,e_begin,e_end}."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    lib = _lib.load()
    n = C.c_int64()
    _lib.check(lib.gm_chunk_table(rp.size - 1, rp.ctypes.data, chunk, int(for_clique), None, 0, C.byref(n)), "gm_chunk_table")
    recs = np.zeros((max(n.value, 1), 4), dtype=np.int32)
    _lib.check(lib.gm_chunk_table(rp.size - 1, rp.ctypes.data, chunk, int(for_clique), recs.ctypes.data, n.value, C.byref(n)),
               "gm_chunk_table")
    return recs[: n.value]


def reduce_scatter_sum(buf, rank: int, world: int):
    """Sum the ranks' equal-sized 1-D integer tensors and return THIS rank's slice [rank * n / world, (rank + 1) * n / world) of the sum:
    `torch.distributed.reduce_scatter_tensor` over RCCL (one ncclReduceScatter over xGMI: every rank receives n / world entries), or --
    gloo, which has no reduce-scatter -- an all-reduce of the host copy followed by the slice (CPU tests, the one-GPU dry run).
    Unsigned 32-bit counters travel as int32: addition wraps the same way."""
    import torch
    import torch.distributed as dist

    n = buf.numel()
    assert n % world == 0, (n, world)
    per = n // world
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return buf[rank * per:(rank + 1) * per]
    if dist.get_backend() == "nccl":
        out = torch.empty(per, dtype=buf.dtype, device=buf.device)
        dist.reduce_scatter_tensor(out, buf)
        return out
    h = buf.cpu()
    dist.all_reduce(h)
    return h[rank * per:(rank + 1) * per].to(buf.device)


def diamond_step(sym, rank: int, world: int, counts, buf=None, stream=0, **kw):
    """One step of the several-rank diamond (the one-GPU algorithm at every N, include/graphminer_amd.h gm_diamond_support_*): this rank's
    share of the triangle pass -> ONE reduce-scatter of the support arrays -> sum C(t, 2) of the rank's slice into counts[0] (an int64
    device tensor) -> the caller all-reduces `counts` as for every other pattern. `buf`: the rank's support array (int32 tensor of
    diamond_support_size entries) to reuse between steps. Returns (buf, slice tensor)."""
    import torch

    from .solvers import diamond_support_finish, diamond_support_partial, diamond_support_size

    n = diamond_support_size(sym, world)
    if buf is None or buf.numel() != n:
        buf = torch.empty(n, dtype=torch.int32, device=counts.device)
    diamond_support_partial(sym, buf.data_ptr(), n, rank=rank, world=world, stream=stream, d_counts=counts.data_ptr(), **kw)
    mine = reduce_scatter_sum(buf, rank, world)
    diamond_support_finish(sym, mine.data_ptr(), mine.numel(), stream=stream, d_counts=counts.data_ptr())
    return buf, mine
