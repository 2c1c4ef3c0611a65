"""The identities behind the hub-corner kernels (graphminer_amd/csrc/gm_ctc.hip), checked on the CPU against the oracle (no GPU):

  * triangle count:  sum over the DAG edges u -> v of |N+(u) ^ N+(v)|  (src/triangle/omp_base.cc:15-21)
      = the same sum over the edges whose source lies BELOW the corner  +  sum_{i<j, M_ij} popc(M_i & M_j)  over the H x H corner M of the
        adjacency matrix of the topologically numbered DAG (every out-neighbour of a corner vertex is a corner vertex);
  * edge supports (diamond, src/sgl/cpu_kernels/diamond.h:1-14: sum_e C(t(e), 2) with t = |N(u) ^ N(v)|): for an edge inside the corner
      t(i, j) = (A A)_ij over the symmetric corner A = M + M^T  +  the common neighbours BELOW the corner.

The kernels evaluate the matrix forms with FP4 MFMA; here they are plain numpy, the rest is the oracle's loop nest restated with sets."""
import numpy as np
import pytest

import oracle as O
from graphminer_amd.rmat import rmat_csr_numpy


def _topological_dag(g):
    """the oriented graph (Graph::orientation's rule, src/common/graph.cc:246-247) renumbered ascending in (degree, id): rows as sets"""
    rp, col = g.row_ptr, g.col_idx
    nv = len(rp) - 1
    deg = np.diff(rp)
    order = np.lexsort((np.arange(nv), deg))
    new = np.empty(nv, np.int64)
    new[order] = np.arange(nv)
    rows = [set() for _ in range(nv)]
    for u in range(nv):
        for v in col[rp[u]:rp[u + 1]]:
            if (deg[u], u) < (deg[v], v):
                rows[new[u]].add(int(new[v]))
    return rows


@pytest.mark.parametrize("scale,ef,h", [(9, 12, 64), (9, 12, 200), (10, 8, 512), (8, 16, 256)])
def test_triangles_and_supports_split_at_the_corner(scale, ef, h):
    g = rmat_csr_numpy(scale, ef, seed=scale * 100 + h)
    osym = O.OGraph(g.row_ptr, g.col_idx)
    odag = O.orient(osym)
    want_tc, want_diamond = O.tc(odag), O.diamond(osym)
    rows = _topological_dag(g)
    nv = len(rows)
    h = min(h, nv)
    base = nv - h
    assert all(min(r) > u for u, r in enumerate(rows) if r), "topological: every edge goes to a larger id"
    # ---- triangle count
    below = sum(len(rows[u] & rows[v]) for u in range(base) for v in rows[u])
    M = np.zeros((h, h), dtype=np.int64)
    for u in range(base, nv):
        assert all(v >= base for v in rows[u])  # the corner is closed under out-edges
        for v in rows[u]:
            M[u - base, v - base] = 1
    assert np.array_equal(M, np.triu(M, 1))
    corner = int((M * (M @ M.T)).sum())  # sum_{i,j} M_ij (M M^T)_ij = sum_{i<j, M_ij} popc(M_i & M_j)
    assert below + corner == want_tc
    # ---- edge supports: t(e) of every DAG edge, the corner's from (A A)_ij + the common in-neighbours below the corner
    A = M + M.T
    AA = A @ A
    inn = [set() for _ in range(nv)]
    for u in range(nv):
        for v in rows[u]:
            inn[v].add(u)
    total = 0
    for u in range(nv):
        for v in rows[u]:
            if u >= base:
                t = int(AA[u - base, v - base]) + len({k for k in inn[u] & inn[v] if k < base})
            else:
                t = len((rows[u] | inn[u]) & (rows[v] | inn[v]))
            total += t * (t - 1) // 2
    assert total == want_diamond
