"""The N>1 path on CPU: 2 (and 3) processes, gloo backend. Each rank takes ITS share of the library's
task-chunk table (gm_chunk_table + gm_partition: the same index arithmetic the GPU launch uses),
computes the partial count of exactly those task edges with the CPU oracle (standing in for the kernel,
which needs a GPU), and the per-rank counts are combined by graphminer_amd.dist.allreduce_counts --
the one collective of the path. The sum must equal the golden count of the reference."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, policy, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    import oracle as O
    from common import load_graph
    from graphminer_amd import dist

    r, w, _ = dist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = load_graph(name)
    dag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    table = dist.chunk_table(dag.row_ptr)
    L = O.lib()
    rp, ci = dag.row_ptr, dag.col_idx
    src = np.repeat(np.arange(dag.nv, dtype=np.int64), np.diff(rp))
    tc = tasks = 0
    for c in dist.partition(len(table), rank, world, policy):
        _, _, eb, ee = (int(x) for x in table[c])
        for e in range(eb, ee):  # task edge (u, v): |N+(u) ^ N+(v)|  (a SPLIT chunk owns part of a row)
            u, v = int(src[e]), int(ci[e])
            a, b = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
            tc += L.gmo_intersect_num(a.ctypes.data if a.size else None, a.size, b.ctypes.data if b.size else None, b.size)
        tasks += ee - eb
    total = dist.allreduce_counts([tc, tasks])
    q.put((rank, tc, total))
    import torch.distributed as td
    td.destroy_process_group()


@pytest.mark.parametrize("world,policy,name", [(2, 0, "rmat10_ef16_s42"), (3, 1, "cora")])
def test_partial_counts_allreduce_to_the_golden_total(world, policy, name):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import GOLDEN

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, policy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, _, total in res:
        assert total == [GOLDEN[name]["tc"], GOLDEN[name]["dag_ne"]]  # every rank holds the reduced total
    assert sum(r[1] for r in res) == GOLDEN[name]["tc"]
    if world > 1 and GOLDEN[name]["tc"] > 0:
        assert all(r[1] < GOLDEN[name]["tc"] for r in res)  # the work really was split


def test_partition_covers_every_chunk_exactly_once():
    sys.path.insert(0, ROOT)
    from graphminer_amd import dist

    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            for policy in (0, 1):
                seen = sorted(c for r in range(world) for c in dist.partition(n, r, world, policy))
                assert seen == list(range(n)), (n, world, policy)


def test_chunk_table_is_a_partition_of_the_edges():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import load_graph
    from graphminer_amd import dist

    for name in ("citeseer", "rmat12_ef8_s7", "rmat14_ef16_s42"):
        g = load_graph(name)  # symmetric graph: rmat14 has rows > 1024 entries -> SPLIT chunks
        for chunk, clique in ((0, False), (64, False), (1024, False), (0, True)):
            t = dist.chunk_table(g.row_ptr, chunk, clique)
            assert t[0, 2] >= 0 and t[-1, 3] == g.E()
            covered = np.zeros(g.E(), dtype=np.int32)
            for ub, ue, eb, ee in t:
                assert ub < ue and eb < ee
                assert g.row_ptr[ub] <= eb and ee <= g.row_ptr[ue]
                covered[eb:ee] += 1
                if clique:
                    assert eb == g.row_ptr[ub] and ee == g.row_ptr[ue]  # whole rows only
            assert np.all(covered == 1)


def _wrap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    from graphminer_amd import dist

    dist.init_from_env("gloo")
    # the shapes the kernels really hand over: rank 0 of gm_motif_formula holds sum C(d,2) - 3T0, every other rank 0 - 3Tr
    # (a value just below 2**64); the pentagon partial is a signed even number; plus a plain count
    wedges_base, t = 10**15, [123456789, 987654321, 55555]
    partial_w = ((wedges_base if rank == 0 else 0) - 3 * t[rank]) % 2**64
    partial_p = (-2 * (rank + 1) if rank else 2 * 10**12) % 2**64
    out = dist.allreduce_counts([partial_w, t[rank], partial_p, 2**64 - 1])
    q.put((rank, [partial_w, partial_p], out))
    import torch.distributed as td
    td.destroy_process_group()


def test_allreduce_counts_is_exact_modulo_2_64():
    """Per-rank partials are defined modulo 2**64 and exceed 2**63: they travel as two's complement."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wrap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    t = [123456789, 987654321, 55555]
    want = [10**15 - 3 * sum(t), sum(t), 2 * 10**12 - 2 * 2 - 2 * 3, (3 * (2**64 - 1)) % 2**64]
    assert any(pw >= 2**63 for _, (pw, _), _ in res)  # the case that used to raise "Overflow when unpacking long long"
    for _, _, out in res:
        assert out == want


def test_allreduce_counts_single_process_keeps_uint64_range():
    sys.path.insert(0, ROOT)
    from graphminer_amd import dist

    assert dist.allreduce_counts([2**64 - 5, 0, 2**63, 7]) == [2**64 - 5, 0, 2**63, 7]


def _diamond_worker(rank, world, port, name, q):
    """the several-rank diamond on CPU: this rank's share of the DAG's task edges (every world-th one) contributes three support increments
    per triangle it finds; reduce_scatter_sum adds the ranks' arrays and hands every rank its slice; sum C(t, 2) of the slice; all-reduce"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    import torch

    import oracle as O
    from common import load_graph
    from graphminer_amd import dist

    dist.init_from_env("gloo")
    g = load_graph(name)
    dag = O.orient(O.OGraph(g.row_ptr, g.col_idx))
    rp, ci = np.asarray(dag.row_ptr), np.asarray(dag.col_idx)
    ne = int(ci.size)
    n = (ne + 64 * world - 1) // (64 * world) * (64 * world)  # gm_diamond_support_size
    sup = np.zeros(n, dtype=np.int64)
    src = np.repeat(np.arange(dag.nv, dtype=np.int64), np.diff(rp))
    for e in range(rank, ne, world):  # task edge (u, v): every common out-neighbour w closes a triangle with the entries u->v, u->w, v->w
        u, v = int(src[e]), int(ci[e])
        a, b = ci[rp[u]:rp[u + 1]], ci[rp[v]:rp[v + 1]]
        common, ia, ib = np.intersect1d(a, b, assume_unique=True, return_indices=True)
        sup[e] += common.size
        np.add.at(sup, rp[u] + ia, 1)
        np.add.at(sup, rp[v] + ib, 1)
    mine = dist.reduce_scatter_sum(torch.from_numpy(sup.astype(np.int32)), rank, world)
    t = mine.to(torch.int64)
    part = int((t * (t - 1) // 2).sum())
    total = dist.allreduce_counts([part, int(sup.sum())])
    q.put((rank, part, total, int(mine.numel()), n))
    import torch.distributed as td
    td.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "citeseer"), (3, "rmat10_ef16_s42")])
def test_diamond_supports_reduce_scatter_to_the_golden_total(world, name):
    """the exchange step of the several-rank diamond (graphminer_amd.dist.reduce_scatter_sum; RCCL on the GPU box, gloo here): per-rank
    support arrays from disjoint task shares -> every rank's slice of their sum -> sum C(t, 2) -> the reference's diamond count"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import GOLDEN

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diamond_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, _, total, per, n in res:
        assert total == [GOLDEN[name]["diamond"], 3 * GOLDEN[name]["tc"]] and per * world == n
    assert sum(r[1] for r in res) == GOLDEN[name]["diamond"]
