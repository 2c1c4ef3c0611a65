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
