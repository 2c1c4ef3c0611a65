#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric ("million edges processed/sec + total match count") on MI355X.

A "step" = one pass of the hot path (the mining kernel(s) of one workload over every task edge of the graph, plus the
RCCL all-reduce of the 64-bit counts when --gpus > 1), inputs resident in HBM before the timed region.

Default run = ALL FIVE BASELINE.json configs on one JSON line:
  * the top-level keys (`value`, `ms_per_step`, `roofline`, `cpu_baseline`, ...) are the HEADLINE: configs[1], triangle
    counting on LiveJournal. The real LiveJournal / Orkut files are not in the image, so the graphs
    are the stand-ins SURVEY.md section 8d names: R-MAT scale 22 ef 10 (TC, diamond), scale 22 ef 28 (4-clique),
    scale 24 ef 16 (3-motif), seed 42; `--data-dir D` runs D/livej/graph.* and D/com-orkut/graph.* instead and checks the
    README known answers;
  * `configs` holds one record per BASELINE config (1: the CPU plumbing case on citeseer, 2: TC, 3: diamond, 4: 4-clique,
    5: 3-motif R-MAT-24 -- the largest single-GPU configuration), each with its count, count check, algorithmic and
    counter-traffic bandwidth, the compulsory floor, and a CPU baseline timed on this box's host cores.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
  python bench.py --workload diamond --scale 20 --ef 16          # one workload (development / profiling)

Multi-GPU: one process per GPU; every rank holds the full CSR (it regenerates / reloads it, no broadcast needed), owns
the task chunks c = rank (mod world) of the cost-ordered dequeue list [Scheduler::round_robin policy,
src/common/scheduler.cc:34-85] and the per-rank counts are summed by ONE all-reduce per step (torch.distributed backend
"nccl" = RCCL over xGMI). Per-GPU work shrinks as N grows: scaling = "strong" (the graph, hence total work, is fixed).

Timing follows src/triangle/gpu_base.cu:53-71 (timer around the kernel loop only, TEPS = nnz / t) with the reference's
averaging (OSDI-experiments-guide.md:39-41: mean of the runs): W untimed steps, then K steps between barriers, max over ranks.
"""
import argparse
import ctypes as C
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (default scale, edge factor, oriented?, description)
    "tc": (22, 10, True, "triangle counting, LiveJournal stand-in"),
    "diamond": (22, 10, False, "sgl diamond, LiveJournal stand-in"),
    "clique4": (22, 28, True, "4-clique, com-Orkut stand-in"),
    "motif3": (24, 16, False, "3-motif, R-MAT scale 24"),
    "motif3e": (24, 16, False, "3-motif, per-edge enumeration (automine_3motif's loop nest), R-MAT scale 24"),
    "rectangle": (16, 16, False, "sgl rectangle (4-cycle), R-MAT scale 16"),
    "house": (16, 16, False, "sgl house, R-MAT scale 16"),
    "pentagon": (16, 16, False, "sgl pentagon, R-MAT scale 16"),
    "clique5": (20, 16, True, "5-clique, R-MAT scale 20"),
    "motif3f": (24, 16, False, "3-motif, formula variant (motif_gpu_formula), R-MAT scale 24"),
    "motif4": (18, 16, False, "4-motif (formula form: per-edge sums + 4-cycles + 4-cliques), R-MAT scale 18"),
}
# BASELINE.json configs[1..4] (configs[0] = the CPU plumbing case, handled by config1_record)
BASELINE_CONFIGS = [
    (2, "tc", "triangle counting on LiveJournal, 1xMI355X, count bit-exact vs tc_omp_base", "livej"),
    (3, "diamond", "sgl diamond pattern on LiveJournal, 1xMI355X (2-level set-intersect DFS)", "livej"),
    (4, "clique4", "4-clique (k-CL k=4) on Orkut, edge-partitioned across 8xMI355X with RCCL count all-reduce", "com-orkut"),
    (5, "motif3", "3-motif counting on a synthetic RMAT scale-24 graph, 1->8 GPU scaling + HBM BW fraction", None),
]
# rocprofv3 kernel names that make up one launch of a workload (everything between the library's two timing events)
KERNELS = {
    "tc": ["tch_kernel", "mine_kernel<0,", "core_tc_"],   # (core_tc_*: the triangles of the hub corner on the matrix cores, gm_ctc.hip; tch_kernel: the shorter list of every edge streamed against a hashed set, gm_tch.hip; mine_kernel<0>: rows beyond its stage)
    # general kernel <P, 0> (+ the sorted-copy classes <P, 1>, <P, 2>), the hashed-row classes and the kernel of the giant rows
    # (one GPU: edge supports from the DAG's triangles, gm_sup.hip; several ranks: the per-edge kernels)
    "diamond": ["core_tc_", "gm::sup_kernel", "gm::sup_near_kernel", "gm::sup_far_kernel", "gm::sup_long_kernel", "gm::sup_pairs_kernel", "mine_kernel<1,", "hrow_kernel<1,", "giant_kernel<1,"],
    # (gm_motif, k = 3: the triangles of the DAG + wedges = sum C(d,2) - 3T; "motif3e": one bounded intersection per edge of the symmetric graph)
    "motif3": ["tch_kernel", "mine_kernel<0,", "core_tc_"],
    "motif3e": ["mine_kernel<2,", "hrow_kernel<2,", "giant_kernel<2,"],
    "clique4": ["cgather_kernel", "cgatherb_kernel", "cbuild_kernel", "clique_mma_kernel", "clique_small_kernel", "mine_kernel<3,"],
    "clique5": ["mine_kernel<4,"],
    "motif3f": ["tch_kernel", "mine_kernel<0,", "core_tc_"],
    "rectangle": ["rect_acc_kernel", "rect_lds_kernel"],
    "house": ["house_acc_kernel", "house_lds_kernel"],
    "pentagon": ["pent_acc_kernel"],
    # (gm_motif, k = 4: per-edge sums of the symmetric graph + rectangle by wedge accumulation + 4-clique of the oriented copy)
    "motif4": ["mine_kernel<5,", "hrow_kernel<5,", "giant_kernel<5,", "rect_acc_kernel", "rect_lds_kernel", "mine_kernel<3,", "cbuild_kernel", "cgather_kernel", "clique_mma_kernel",
               "clique_small_kernel", "cgatherb_kernel", "tch_kernel", "core_tc_"],
}
CORNER_KERNEL = "core_tc_"  # (the MFMA kernels of gm_ctc.hip: reported beside the whole-launch HBM roofline against the FP4 peak)
TRAFFIC_MARKER = "issue_calib_kernel"  # the dispatch in front of every workload of the traffic worker (measure_traffic)
DIAMOND_SUPPORTS_MAX_WORLD = int(os.environ.get("GM_DIAMOND_SUPPORTS_MAX_WORLD", "4"))  # (graphminer_amd/host/multi.cc has the same rule)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable by a stream)
M64 = (1 << 64) - 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="", choices=[""] + sorted(WORKLOADS), help="run ONE workload instead of the five BASELINE configs")
    ap.add_argument("--configs", default="all", help="'all' or a comma list out of tc,diamond,clique4,motif3 (default run)")
    ap.add_argument("--scale", type=int, default=0)
    ap.add_argument("--ef", type=int, default=0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--graph", default="", help="prefix of graph.meta.txt/.vertex.bin/.edge.bin (real dataset), one-workload mode")
    ap.add_argument("--data-dir", default=os.environ.get("GM_DATA_DIR", ""), help="directory with livej/graph.* and com-orkut/graph.* (real datasets)")
    ap.add_argument("--uniform", default="", help="NV,M: uniform random graph instead of R-MAT (LiveJournal-size flat-degree stand-in)")
    ap.add_argument("--powerlaw", default="", help="NV,M,MAXDEG: Chung-Lu power-law graph (LiveJournal: 4847571,43000000,20000)")
    ap.add_argument("--community", default="", help="NV,M: planted communities + power-law background with LiveJournal's triangle density (4847571,43000000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-standins", action="store_true", help="skip the extra LiveJournal-size stand-in graphs of the default run")
    ap.add_argument("--no-ref-baseline", action="store_true", help="skip the timed runs of the oracle/_ref reference binaries")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle's (secondary) TC sample")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off", "file"],
                    help="HBM-side bytes per launch: auto = re-run the workloads under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "(two extra passes, N = 1 only), file = profiles/traffic.json, off = null")
    ap.add_argument("--traffic-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--share-rank", type=int, default=0, help=argparse.SUPPRESS)   # traffic worker at N > 1: measure the share of this rank ...
    ap.add_argument("--share-world", type=int, default=1, help=argparse.SUPPRESS)  # ... of this world on ONE GPU (rank / world are launch arguments)
    ap.add_argument("--tune", default="", help="comma separated gm_launch.tune[] override")
    ap.add_argument("--policy", type=int, default=0, help="0 = chunked round robin, 1 = contiguous ranges")
    ap.add_argument("--first-call-repeats", type=int, default=2, help="extra FRESH handles over the same device arrays whose first call (setup + one count) is timed "
                    "beside the first one (N = 1): first_call_ms = the median, every run listed")
    ap.add_argument("--detail", default="", help="where the FULL record goes (default gpurun_out/bench_detail.json); stdout carries one compact line")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# graphs
# ---------------------------------------------------------------------------------------------------------------
_T0 = time.perf_counter()


def progress(msg):
    """phase marks on stderr (rank 0): where a long run is, and where a stuck one stopped"""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def flush_c_stdio():
    """fflush(NULL): whatever native libraries left in the C stdio buffers of this process goes out now"""
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


class BenchGraph:
    """symmetric CSR resident in HBM (+ torch views of its arrays) and, on demand, its oriented copy"""

    def __init__(self, sym, rp, ci, name, build_s):
        self.sym, self.rp, self.ci, self.name, self.build_s = sym, rp, ci, name, build_s
        self._dag = None

    def dag(self):
        if self._dag is None:
            self._dag = self.sym.orient()
        return self._dag

    def free(self):
        if self._dag is not None:
            self._dag.free()
        self.sym.free()
        self.rp = self.ci = None


def build_graph(a, local_rank, scale, ef, graph_prefix=""):
    import torch

    from graphminer_amd import DeviceGraph, Graph
    from graphminer_amd.rmat import community_csr_device, powerlaw_csr_device, rmat_csr_device, uniform_csr_device

    t = time.perf_counter()
    if graph_prefix:
        h = Graph(graph_prefix)
        dev = torch.device("cuda", local_rank)
        rp = torch.from_numpy(h.row_ptr).to(dev)
        ci = torch.from_numpy(h.col_idx).to(dev)
        sym = DeviceGraph.from_device_ptrs(h.V(), h.E(), rp.data_ptr(), ci.data_ptr(), local_rank, keepalive=(rp, ci))
        name = f"file:{graph_prefix}"
    elif getattr(a, "community", ""):
        cnv, cm = (int(x) for x in a.community.split(","))
        sym, rp, ci = community_csr_device(cnv, cm, 64, 0.6, 20000, a.seed, local_rank)
        name = f"community_nv{cnv}_m{cm}_c64_p0.6_seed{a.seed}"
    elif a.powerlaw:
        pnv, pm, pmax = (int(x) for x in a.powerlaw.split(","))
        sym, rp, ci = powerlaw_csr_device(pnv, pm, pmax, 2.5, a.seed, local_rank)
        name = f"powerlaw_nv{pnv}_m{pm}_max{pmax}_seed{a.seed}"
    elif a.uniform:
        unv, um = (int(x) for x in a.uniform.split(","))
        sym, rp, ci = uniform_csr_device(unv, um, a.seed, local_rank)
        name = f"uniform_nv{unv}_m{um}_seed{a.seed}"
    else:
        sym, rp, ci = rmat_csr_device(scale, ef, a.seed, local_rank)
        name = f"rmat_s{scale}_ef{ef}_seed{a.seed}"
    torch.cuda.synchronize()
    return BenchGraph(sym, rp, ci, name, time.perf_counter() - t)


def alg_bytes_device(workload, bg, lib, g):
    """SURVEY.md 8(d) ALGORITHMIC bytes of one launch over the whole graph, exact, computed on the GPU with torch
    (tests/test_gpu_bench.py checks it against the oracle's gmo_alg_bytes_* on small graphs).
    Returns (bytes or None, compulsory floor 8(nv+1)+4ne of the CSR the workload runs on)."""
    import torch

    from graphminer_amd import _lib

    rp, ci = bg.rp, bg.ci
    nv = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    if workload in ("tc", "clique4", "clique5", "motif3f"):
        # oriented degrees from the symmetric arrays: keep (s -> d) iff deg[d] > deg[s] or (== and d > s)  (graph.cc:246-247)
        src = torch.repeat_interleave(torch.arange(nv, device=rp.device), deg)
        dst = ci.long()
        keep = (deg[dst] > deg[src]) | ((deg[dst] == deg[src]) & (dst > src))
        s2, d2 = src[keep], dst[keep]
        del src, dst, keep
        dplus = torch.bincount(s2, minlength=nv)
        ne = int(s2.numel())
        sq = int((dplus * dplus).sum().item())
        dv = int(dplus[d2].sum().item())
        del s2, d2, dplus
        floor = 8 * (nv + 1) + 4 * ne
        if workload == "clique5":
            return None, floor
        ab = 4 * (sq + dv) + 40 * ne  # TC (motif3f = the TC kernel on the DAG)
        if workload == "clique4":  # level 1 = the TC formula, level 2 from the statistics kernel
            l2 = C.c_uint64(0)
            _lib.check(lib.gm_clique4_level2_bytes(g.handle, C.byref(l2)), "gm_clique4_level2_bytes")
            ab += int(l2.value)
        return ab, floor
    ne = int(ci.numel())
    floor = 8 * (nv + 1) + 4 * ne
    sq = int((deg * deg).sum().item())  # sum_e d(src)
    dv = int(deg[ci.long()].sum().item())  # sum_e d(dst)
    if workload == "diamond":  # edges v1 < v0 only: by symmetry exactly half of sum_e (d(v0)+d(v1))
        return 4 * (sq + dv) // 2 + 40 * (ne // 2), floor
    if workload in ("motif3", "motif3e"):  # one difference per directed edge + one intersect per v1 < v0 edge
        return 4 * (sq + dv) + 4 * (sq + dv) // 2 + 40 * ne, floor
    return None, floor


def kernel_constants():
    """the kernel constants the own-byte model depends on, read from the library (gm_constant: the headers the kernels are compiled with)"""
    from graphminer_amd import _lib

    lib = _lib.load()
    out = {}
    for name in ("tct_stage_max", "topo_min_mean_row", "motif_trim_min_list", "cb_min_deg", "cb_max_deg", "core_h_default", "wide_min_words", "long_list"):
        v = C.c_int64(0)
        _lib.check(lib.gm_constant(name.encode(), C.byref(v)), "gm_constant " + name)
        out[name] = int(v.value)
    return out


def tc_core(bg, workload):
    """the hub corner the triangle count of this graph takes on the matrix cores (gm_tc_core_info, after a launch): tc -> the DAG handle,
    motif3 -> the symmetric handle (its cached oriented copy)"""
    try:
        from graphminer_amd.solvers import tc_core_info

        if workload == "tc":
            return tc_core_info(bg.dag())
        if workload in ("motif3", "motif3f"):
            return tc_core_info(bg.sym)
    except Exception:
        pass
    return None


def own_bytes_device(workload, bg, world=1, core_override=None):
    """OWN-ALGORITHM bytes of one launch over the whole graph: what THIS library's kernels must move by
    construction -- the keys they stream, the task descriptors, every row staged / hashed once, the offsets, the k-clique arena
    written and read once -- exact, with torch on the GPU. `roofline.achieved` = this / kernel time, so `frac` <= 1 by construction
    (SURVEY 8(d)'s figure, which charges the reference's loop nest, is kept beside it as `algorithmic_*`).
      tc       4*sum_e min(d+(v), tail_u(v)) + 12|E+| + 8(nv+1)       (the host of an edge u -> v = the endpoint whose list is NOT streamed: N+(v)
               whole, or the tail of N+(u) beyond v -- the library numbers the DAG topologically -- whichever is shorter; a row > 2048 entries hosts nothing)
      diamond  one GPU (edge supports from the DAG's triangles, gm_sup.hip): the tc figure + 20|E+| + 4 A + 16 W + 8|E+|  (4 B per task for its own
               entry, the support array zeroed, read once, and added to once per task and once per staged entry: 4 x 4|E+|; A = streamed edges
               reported by a 4-byte atomic, W = 64-bit words of the match masks the other streamed edges are reported through, written and
               read once, 8 B of mask offsets per entry; A and W from the library after a launch, gm_diamond_support_info; no masks: A = T)
               several ranks (one intersection per edge): 4*sum_{undirected e} min(d(u), d(v)) + 12*ne + 8(nv+1)
      motif3   the tc figure (gm_motif k = 3 counts the triangles of the DAG; wedges = sum C(d,2) - 3T)
      motif3e  (per-edge enumeration) diamond's several-ranks figure on the copy numbered by DESCENDING degree, the streamed list trimmed to its
               keys < max(u, v) -- its neighbours of higher degree -- when it has >= 128 keys
      clique4  4*K + 16*tasks + 4|E+| + 16(nv+1) + 8*arena words + 4*gathered words   (rows of the WIDE vertices -- d+ > 256 -- whose first
               endpoint lies in the hub core are gathered from the dense core bitmap: the distinct words that hold the probed bits; every other edge
               of an owner is a streamed task, K as for tc with the stage limit 2048; rows with d+ < 3 own no matrix)
    Returns {"bytes", "streamed_keys", "parts"} or None."""
    import torch

    kc = kernel_constants()
    TCT_STAGE_MAX, TOPO_MIN_MEAN_ROW, TRIM_MIN_LIST = kc["tct_stage_max"], float(kc["topo_min_mean_row"]), kc["motif_trim_min_list"]
    CB_MIN_DEG, CB_MAX_DEG = kc["cb_min_deg"], kc["cb_max_deg"]
    rp, ci = bg.rp, bg.ci
    nv = rp.numel() - 1
    deg = rp[1:] - rp[:-1]
    src = torch.repeat_interleave(torch.arange(nv, device=rp.device), deg)
    dst = ci.long()
    if workload in ("tc", "motif3", "motif3f", "clique4"):  # (motif3 = the triangle kernel on the DAG + a closed form)
        keep = (deg[dst] > deg[src]) | ((deg[dst] == deg[src]) & (dst > src))  # graph.cc:246-247
        s2, d2 = src[keep], dst[keep]
        del src, dst, keep
        dplus = torch.bincount(s2, minlength=nv)
        ne = int(s2.numel())
        du, dv = dplus[s2], dplus[d2]
        fixed = 12 * ne + 8 * (nv + 1)
        # position of v in N+(u) under the TOPOLOGICAL numbering the library gives the DAG (ids ascending in (degree, id), gm_graph.hip
        # get_relabeled mode 2): rank the vertices, sort the edges by (new u, new v). An in-edge task (the target hosts) streams only
        # the part of N+(u) beyond v.
        newid = torch.empty(nv, dtype=torch.long, device=rp.device)
        newid[torch.argsort(deg * (1 << 32) + torch.arange(nv, device=rp.device))] = torch.arange(nv, device=rp.device)
        new_u, new_v = newid[s2], newid[d2]
        perm = torch.argsort(new_u * (1 << 32) + new_v)
        ru = new_u[perm]  # the new row of every edge, in (new u, new v) order: rows are contiguous blocks
        pos = torch.empty(ne, dtype=torch.long, device=rp.device)
        pos[perm] = torch.arange(ne, device=rp.device) - torch.searchsorted(ru, ru)
        del newid, perm, ru
        # (the library renumbers -- and trims -- only where lists are long: sum d+^2 / |E+| >= 64, gm_launch.hip topo_view)
        trimmed = ne > 0 and float((dplus * dplus).sum().item()) / ne >= TOPO_MIN_MEAN_ROW
        tail = (du - pos - 1) if trimmed else du  # what an in-edge task streams: N+(u) beyond v, or the whole list
        if workload != "clique4":
            # the host = the endpoint whose list is NOT streamed: N+(v) whole or the tail of N+(u), whichever is shorter (ties: u hosts);
            # a row beyond the stage hosts nothing, and its own out-edges stream N+(v) on the chunked kernel
            u_hosts = (dv > TCT_STAGE_MAX) | (tail >= dv)
            streamed = torch.where(du > TCT_STAGE_MAX, dv, torch.where(u_hosts, dv, tail))
            # the hub corner (gm_ctc.hip): the out-edges of the last H vertices of the renumbered DAG are no tasks of the stream -- one masked
            # bit-matrix product instead: every 256 x 256 block pair IB <= JB stages 2 x 256 rows x 64 B per 512-column chunk from JB / 2 on
            core = core_override if core_override is not None else tc_core(bg, workload)
            corner_bytes, corner_keys = 0, 0
            if core and core["h"] > 0:
                in_corner = new_u >= nv - core["h"]
                corner_keys = int(streamed[in_corner].sum().item())
                streamed = torch.where(in_corner, torch.zeros_like(streamed), streamed)
                if core.get("whole_column_range"):  # the edge supports: (A A)_ij over every column chunk, + the positions and the adds
                    nb, nc = core["h"] // 256, core["h"] // 512
                    corner_bytes = nb * (nb + 1) // 2 * nc * 32768 + 10 * core["edges"]
                elif core["h"] % 512 == 0:
                    nb, nc = core["h"] // 256, core["h"] // 512
                    corner_bytes = sum((jb + 1) * (nc - jb // 2) for jb in range(nb)) * 32768
                else:
                    nj, wt = (core["h"] + 63) // 64, (core["h"] + 31) // 32
                    corner_bytes = sum((jb + 1) * max((wt + 1) // 2 - jb, 0) for jb in range(nj)) * 1024
            k = int(streamed.sum().item())
            parts = {"streamed_keys_x4": 4 * k, "task_descriptors_and_rows_12_per_edge": 12 * ne, "offsets": 8 * (nv + 1)}
            if core and core["h"] > 0:
                parts.update({"corner_operand_chunks": corner_bytes, "corner_h": core["h"], "corner_edges": core["edges"],
                              "keys_the_corner_edges_would_stream": corner_keys})
            return {"bytes": 4 * k + fixed + corner_bytes, "streamed_keys": k, "parts": parts}
        # 4-clique (gm_cbuild.hip / gm_cgather.hip / gm_cmma.hip): u OWNS a bit-matrix when 3 <= d+(u) <= 2048.
        #  * a WIDE owner (matrix > 2048 words: d+ > 256) on a topologically numbered DAG takes its rows whose first endpoint lies in the hub
        #    core (the last core_h ids) by GATHER from the dense core bitmap: bytes by construction = the distinct 4-byte words that hold
        #    the probed bits of a row (positions of N+(u) beyond v);
        #  * every other edge u -> v of an owner is a streamed task hosted like TC's (stage limit 2048), 16 B of task record each;
        #  * every matrix is written once and read once (8 B per arena word), the rows of the hosts are staged once, two offset arrays twice.
        owner = (du >= CB_MIN_DEG) & (du <= CB_MAX_DEG)
        wide = owner & (du * ((du + 31) // 32) > kc["wide_min_words"])
        core_h = min(nv, kc["core_h_default"]) if (trimmed and nv >= 64) else 0
        in_core = wide & (new_v >= nv - core_h) if core_h else torch.zeros_like(owner)
        stream_t = owner & ~in_core
        v_hosts = (dv > tail) & (dv <= CB_MAX_DEG)
        streamed = torch.where(stream_t, torch.where(v_hosts, tail, dv), torch.zeros_like(dv))
        k = int(streamed.sum().item())
        tasks = int(stream_t.sum().item())
        gather_words = 0
        if core_h and bool(in_core.any()):
            # edges in (new u, new v) order: rows are contiguous; word of a column = (new v - base) >> 5; a row's gather touches the distinct
            # words among the columns BEYOND it
            o = torch.argsort(new_u * (1 << 32) + new_v)
            ru_s, w_s, core_s = new_u[o], (new_v[o] - (nv - core_h)) >> 5, in_core[o]
            first = torch.ones(ne, dtype=torch.bool, device=rp.device)
            first[1:] = (ru_s[1:] != ru_s[:-1]) | (w_s[1:] != w_s[:-1])  # a new (row, word) starts here
            cf = torch.cumsum(first.long(), 0)
            row_end = torch.searchsorted(ru_s, ru_s, right=True) - 1  # last edge of the row of every edge
            idx = torch.arange(ne, device=rp.device)
            has_next = idx < row_end
            nxt = torch.clamp(idx + 1, max=ne - 1)
            # distinct words among the edges idx + 1 .. row_end: the (row, word) starts in that range, + 1 when edge idx + 1 continues idx's word
            dist = torch.where(has_next, cf[row_end] - cf[nxt] + 1, torch.zeros_like(cf))
            gather_words = int(dist[core_s].sum().item())
        # round 6: the BLOCKED gather (gm_cgather.hip cgatherb_kernel) reads column tables instead of core rows -- its own bytes are the
        # library's figure for the plan it built (records + row positions + 2 B per column from a unit's first row on + one 64 KB block image
        # per work item); the row-major figure stays in the parts for comparison
        gather_bytes, blocked = 4 * gather_words, None
        try:
            from graphminer_amd import CliqueSolver, _lib as _l

            dagc = bg.dag()
            CliqueSolver(dagc, 4)
            gi = (C.c_int64 * 4)()
            _l.check(_l.load().gm_clique4_gather_info(dagc.handle, gi), "gm_clique4_gather_info")
            if int(gi[0]) > 0:
                blocked = {"units": int(gi[0]), "unit_bytes": int(gi[1]), "items": int(gi[2]), "blocks": int(gi[3])}
                gather_bytes = int(gi[1]) + int(gi[2]) * 65536
        except Exception:
            pass
        dw = dplus[(dplus >= CB_MIN_DEG) & (dplus <= CB_MAX_DEG)]
        arena_words = int((dw * ((dw + 31) // 32)).sum().item())
        return {"bytes": 4 * k + 16 * tasks + 4 * ne + 16 * (nv + 1) + 8 * arena_words + gather_bytes, "streamed_keys": k,
                "parts": {"streamed_keys_x4": 4 * k, "task_records_x16": 16 * tasks, "rows_staged_once": 4 * ne, "offsets": 16 * (nv + 1),
                          "arena_words_written_and_read_x8": 8 * arena_words, "core_words_gathered_x4_row_major": 4 * gather_words,
                          "blocked_gather": blocked, "gather_bytes": gather_bytes,
                          "core_rows_gathered": int(in_core.sum().item()), "core_h": core_h}}
    if workload == "diamond" and world <= 1:
        dag = bg.dag()
        if dag.get_max_degree() <= TCT_STAGE_MAX:  # (longer DAG rows: the library takes the per-edge kernels, below)
            from graphminer_amd import SglSolver, TCSolver, _lib

            del src, dst
            SglSolver(bg.sym, "diamond")
            cinfo = (C.c_int64 * 4)()
            _lib.check(_lib.load().gm_sup_core_info(bg.sym.handle, cinfo), "gm_sup_core_info")
            tcp = own_bytes_device("tc", bg, core_override={"h": int(cinfo[0]), "edges": int(cinfo[1]), "whole_column_range": True})
            tri = int(TCSolver(dag))
            nd = int(dag.E())
            # the streamed edges: match masks where the library uses them (round 5, gm_sup.hip: in-edge tasks store which keys of their tail
            # matched, two kernels sum the masks by column) -- the mask arena written and read once (16 B per 64-bit word), 8 B per entry for
            # the two offset arrays -- and one 4-byte atomic for every other streamed edge; both figures are the library's own
            # (gm_diamond_support_info after a launch).  Without masks: one atomic per triangle.
            SglSolver(bg.sym, "diamond")
            info = (C.c_int64 * 4)()
            _lib.check(_lib.load().gm_diamond_support_info(bg.sym.handle, info), "gm_diamond_support_info")
            mask_words, atomics = int(info[0]), int(info[1])
            if not mask_words:
                atomics = tri
            extra = 16 * mask_words + (8 * nd if mask_words else 0)
            return {"bytes": tcp["bytes"] + 20 * nd + 4 * atomics + extra, "streamed_keys": tcp["streamed_keys"],
                    "parts": dict(tcp["parts"], own_entry_per_task_x4=4 * nd, supports_zeroed_read_added_x16=16 * nd, streamed_edge_atomics_x4=4 * atomics,
                                  match_mask_words_written_and_read_x16=16 * mask_words, mask_offsets_per_entry_x8=8 * nd if mask_words else 0,
                                  triangles=tri, shortest_masked_tail=int(info[3]))}
    if workload in ("diamond", "motif3e"):
        ne = int(ci.numel())
        if workload == "motif3e":
            # the enumeration runs on the library's copy numbered by DESCENDING degree (gm_motif, gm_graph.hip get_relabeled mode 1: new id =
            # nv - 1 - rank in ascending (degree, id)): "below max(u, v)" then means "of higher degree", and the trimmed lists are short
            rank_ = torch.empty(nv, dtype=torch.long, device=rp.device)
            rank_[torch.argsort(deg * (1 << 32) + torch.arange(nv, device=rp.device))] = torch.arange(nv, device=rp.device)
            newid = nv - 1 - rank_
            src, dst = newid[src], newid[dst]
            deg = torch.bincount(src, minlength=nv)  # degrees by new id
            rp = torch.cat([torch.zeros(1, dtype=deg.dtype, device=deg.device), torch.cumsum(deg, 0)])
            order_ = torch.argsort(src * (1 << 32) + dst)
            src, dst = src[order_], dst[order_]
            ci = dst
            del rank_, newid, order_
        und = dst < src  # every undirected edge once
        u, v = src[und], dst[und]
        del src, dst, und
        a, b = deg[u], deg[v]
        u_longer = (a > b) | ((a == b) & (u > v))  # sym_hosts (gm_mine.h): the longer row hosts, the shorter list is streamed
        short = torch.where(u_longer, v, u)
        slen = torch.where(u_longer, b, a)
        if workload == "motif3e":
            hi = torch.maximum(u, v)
            keys = torch.repeat_interleave(torch.arange(nv, device=rp.device), deg) * (1 << 32) + ci.long()  # the CSR as sorted (row, id) keys
            below = torch.searchsorted(keys, short * (1 << 32) + hi) - rp[short]  # keys of N(short) below max(u, v)
            del keys, hi
            slen = torch.where(slen >= TRIM_MIN_LIST, below, slen)
        k = int(slen.sum().item())
        return {"bytes": 4 * k + 12 * ne + 8 * (nv + 1), "streamed_keys": k,
                "parts": {"streamed_keys_x4": 4 * k, "entries_and_descriptors_12_per_entry": 12 * ne, "offsets": 8 * (nv + 1)}}
    return None


# ---------------------------------------------------------------------------------------------------------------
# one workload: W warm-up steps, K timed steps
# ---------------------------------------------------------------------------------------------------------------
class Runner:
    def __init__(self, a):
        import torch
        import torch.distributed as dist

        self.a, self.torch, self.dist = a, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path to measure)"
        # GM_BENCH_ONE_GPU=1 (scripts/scale_dryrun.sh): every rank of a several-rank job on GPU 0 -- the N > 1 code path (shares, collective,
        # max-over-ranks timing, rank-0-only reporting with the others parked at the barrier) on a one-GPU box, over gloo
        self.one_gpu = os.environ.get("GM_BENCH_ONE_GPU") == "1"
        if self.one_gpu:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.use_dist = self.world > 1 or os.environ.get("GM_BENCH_FORCE_DIST") == "1"  # the latter: exercise RCCL with one rank
        # "nccl" = RCCL over xGMI; "gloo": the dry run (counts staged through the host) -- the default when every rank shares GPU 0,
        # where RCCL refuses two ranks on one device
        self.backend = os.environ.get("GM_BENCH_BACKEND", "gloo" if (self.one_gpu and self.world > 1) else "nccl")
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            import datetime

            # (rank 0 reports alone at the end -- CPU baselines, rocprofv3 passes -- while the others wait in the closing barrier)
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev, timeout=datetime.timedelta(minutes=60))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, timeout=datetime.timedelta(minutes=60))
        # the ranks the collective actually reaches: an all-reduce of ones (1 without a process group)
        self.ranks_seen = 1
        if self.use_dist:
            ones = torch.ones(1, dtype=torch.int64, device=self.dev)
            self.all_reduce(ones)
            self.ranks_seen = int(ones.item())
            assert self.ranks_seen == self.world, f"the all-reduce saw {self.ranks_seen} ranks of {self.world}"
            # RCCL prints a version banner (RCCL / HIP / ROCm version, hostname, library path) through C stdio when its first communicator
            # comes up; on a pipe that buffer is flushed at process exit -- AFTER the JSON line.  Flush it now, on every rank: the JSON
            # line must be the last thing on stdout.
            flush_c_stdio()
        from graphminer_amd import _lib

        self._lib = _lib
        self.lib = _lib.load()
        self.counts = torch.zeros(8, dtype=torch.int64, device=self.dev)
        # the share of the task chunks this process launches: its rank of the job, or (traffic worker) a named share on one GPU
        self.launch_rank, self.launch_world = (self.rank, self.world) if self.world > 1 else (a.share_rank, max(a.share_world, 1))

    def all_reduce(self, t, op=None):
        """the job's collective: RCCL on the device tensor, or -- gloo dry run -- the same reduction staged through the host"""
        op = op if op is not None else self.dist.ReduceOp.SUM
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if self.use_dist:
            flush_c_stdio()
            self.dist.barrier()  # rank 0 finishes its host-side reporting before any rank tears the communicator down
            self.dist.destroy_process_group()
            flush_c_stdio()

    def run(self, workload, bg, steps, warmup, solo=False):
        """Returns a dict with the timing and the counts of `workload` on graph `bg`. solo: this process alone, the whole graph, no
        collective (rank 0's CPU-baseline comparison runs while the other ranks wait)."""
        from graphminer_amd._lib import gm_launch, gm_stats

        a, lib, torch = self.a, self.lib, self.torch
        oriented = WORKLOADS[workload][2]
        g = bg.dag() if oriented else bg.sym
        la = gm_launch()
        la.stream = torch.cuda.current_stream().cuda_stream or None
        la.rank, la.world, la.policy = (0, 1, a.policy) if solo else (self.launch_rank, self.launch_world, a.policy)
        use_dist = self.use_dist and not solo

        def fence():
            if use_dist:
                self.fence()
            else:
                torch.cuda.synchronize()
        la.d_counts = self.counts.data_ptr()
        if a.tune:
            for i, t in enumerate(a.tune.split(",")):
                la.tune[i] = int(t)
        st = gm_stats()

        # diamond on several ranks: the one-GPU algorithm at every N (graphminer_amd.dist.diamond_step: a share of the triangle pass, ONE
        # reduce-scatter of the support arrays over xGMI, sum C(t, 2) of the rank's slice) -- unless the per-edge kernels are asked for
        # (--tune ... 0x10000000) or a row of the oriented copy exceeds the stage (GM_ERR_UNSUPPORTED at the first call)
        # ... up to DIAMOND_SUPPORTS_MAX_WORLD ranks: beyond, a rank's share of the per-edge kernels + the single 8-byte all-reduce north_star
        # names is the better deal (one-GPU simulation of the shares, profiles/r05/sim_scale_one_gpu.txt, R-MAT-22 at 2 / 4 / 8 ranks: shared
        # triangle pass 4.48 / 3.04 / 2.35 ms + a reduce-scatter of 81 / 122 / 142 MB per rank, per-edge kernels 8.20 / 4.75 / 2.90 ms + 8 bytes)
        dstate = {"on": workload == "diamond" and 1 < la.world <= DIAMOND_SUPPORTS_MAX_WORLD and not (la.tune[6] & 0x10000000),
                  "buf": None}

        def diamond_sup_step():
            from graphminer_amd import dist as gdist

            buf = dstate["buf"]
            # (RCCL: reduce_scatter_tensor; the gloo dry run stages the exchange through the host, graphminer_amd.dist.reduce_scatter_sum)
            # (a named share on one GPU -- the traffic worker -- runs the same two launches on its slice, without the collectives)
            dstate["buf"], _ = gdist.diamond_step(g, la.rank, la.world, self.counts, buf=buf, stream=la.stream or 0, tune=list(la.tune))
            if use_dist:
                self.all_reduce(self.counts)

        def step(h=None):
            if dstate["on"]:
                try:
                    return diamond_sup_step()
                except self._lib.GraphMinerError as e:
                    if e.status != self._lib.GM_ERR_UNSUPPORTED:
                        raise
                    dstate["on"] = False
            hd = (h or g).handle
            if workload == "tc":
                rc = lib.gm_tc(hd, C.byref(la), None, C.byref(st))
            elif workload in ("diamond", "rectangle", "house", "pentagon"):
                rc = lib.gm_sgl(hd, workload.encode(), C.byref(la), None, C.byref(st))
            elif workload in ("clique4", "clique5"):
                rc = lib.gm_clique(hd, int(workload[-1]), C.byref(la), None, C.byref(st))
            elif workload == "motif3f":
                rc = lib.gm_motif_formula(hd, 3, C.byref(la), None, 2, C.byref(st))
            elif workload == "motif4":
                rc = lib.gm_motif(hd, 4, C.byref(la), None, 6, C.byref(st))
            elif workload == "motif3e":
                la.tune[6] |= 0x10000000
                rc = lib.gm_motif(hd, 3, C.byref(la), None, 2, C.byref(st))
            else:
                rc = lib.gm_motif(hd, 3, C.byref(la), None, 2, C.byref(st))
            self._lib.check(rc, "bench step")
            if use_dist:
                self.all_reduce(self.counts)  # ONE RCCL all-reduce of the 64-bit counts (int64 add wraps like uint64)

        # first call: builds the per-graph task tables ("Time on generating the edgelist" of the reference) -- timed apart
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        first_call_s = time.perf_counter() - t0
        for _ in range(max(warmup - 1, 0)):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            self.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        result = [int(x) & M64 for x in self.counts.cpu().tolist()]
        # The first call again, on FRESH handles over the same device arrays (the driver's run of round 4 saw 441 ms where
        # every builder run saw 80): a fresh handle rebuilds everything -- orientation, renumbered copy, task lists, key stream, tables --
        # so one stalled run shows as the outlier it is.  first_call_ms = the median of the runs; all of them are listed.
        first_runs = [{"first_call_ms": 1e3 * first_call_s, "end_to_end_ms": None}]
        if self.world == 1 and not solo and not dstate["on"] and a.first_call_repeats > 0 and self.launch_world == 1:
            from graphminer_amd import DeviceGraph

            for _ in range(a.first_call_repeats):
                sym2 = DeviceGraph.from_device_ptrs(bg.sym.V(), bg.sym.E(), bg.rp.data_ptr(), bg.ci.data_ptr(), self.local_rank, keepalive=(bg.rp, bg.ci))
                try:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    g2 = sym2.orient() if oriented else sym2
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    step(g2)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    again = [int(x) & M64 for x in self.counts.cpu().tolist()]
                    assert again[:2] == result[:2], (again[:2], result[:2])
                    first_runs.append({"first_call_ms": 1e3 * (t2 - t1), "end_to_end_ms": 1e3 * (t2 - t0), "setup_ms": g2.setup_times_ms()})
                finally:
                    if oriented and "g2" in locals() and g2 is not sym2:
                        g2.free()
                    sym2.free()
        if dstate["on"]:  # two launches per step (the share of the triangle pass, sum C(t, 2) of the slice): their sum is the step's kernel time
            kms = g.kernel_times_ms(min(2 * steps, 64))
            k_avg = sum(kms) / max(len(kms) // 2, 1)
        else:
            kms = g.kernel_times_ms(min(steps, 64))
            k_avg = sum(kms) / max(len(kms), 1)
        cms = g.corner_times_ms(min(2 * steps if dstate["on"] else steps, 64))
        corner_avg = sum(cms) / max(len(cms) // (2 if dstate["on"] else 1), 1)
        per_gpu = [k_avg]
        if use_dist:  # per-GPU kernel time, as the reference prints runtime[gpu i] (src/clique/multigpu.cu:136-137)
            tk = torch.zeros(self.world, dtype=torch.float64, device=self.dev)
            tk[self.rank] = k_avg
            self.all_reduce(tk)
            per_gpu = [float(x) for x in tk.cpu().tolist()]
        sym_e = bg.sym.E()
        # "edges processed" = the reference's nnz (src/triangle/gpu_base.cu:69): |E+| (tc, clique), ne/2 (diamond), ne (motif)
        tasks = g.E() // 2 if workload in ("diamond", "rectangle", "house", "pentagon") else g.E()
        setup = g.setup_times_ms()  # (the DAG handle carries the orientation time of the graph it was made from)
        return {
            "workload": workload, "g": g, "tasks": tasks, "elapsed": elapsed, "steps": steps,
            "ms_per_step": 1e3 * elapsed / steps, "count": result[:2] if workload.startswith("motif3") else (result[:6] if workload == "motif4" else result[0]),
            "kernel_ms_avg": k_avg, "corner_ms_avg": corner_avg, "per_gpu_kernel_ms": per_gpu, "first_call_ms": median([x["first_call_ms"] for x in first_runs]), "setup_ms": setup,
            "first_call_runs_ms": [round(x["first_call_ms"], 2) for x in first_runs],
            # graph resident in HBM -> first count: the first call (builds tables, renumbered copies, task lists, runs once) + the orientation
            # of a DAG workload, which bench.py asks for before the call (the symmetric-graph solvers orient inside their first call)
            "end_to_end_ms": 1e3 * first_call_s + (float(setup.get("orient_ms", 0.0)) if oriented else 0.0),
            # ONE SHOT, as `tc_gpu_base <graph>` is used: a fresh handle over the resident CSR -> orientation -> every table ->
            # one count, wall clock, the median of the fresh-handle runs (None when none was made: several ranks, --first-call-repeats 0)
            "single_shot_ms": (median([x["end_to_end_ms"] for x in first_runs if x.get("end_to_end_ms") is not None])
                               if any(x.get("end_to_end_ms") is not None for x in first_runs) else None),
            "nv": g.V(), "ne_sym": sym_e, "max_degree": g.get_max_degree(), "stats": {"grid": int(st.grid), "block": int(st.block)},
            "diamond_supports_across_ranks": bool(dstate["on"]),
        }

    def stream_ceiling(self):
        """Measured stream read bandwidth (gm_stream_ceiling: 16 B per lane over 4 GiB >> the 256 MiB Infinity Cache), GB/s."""
        torch = self.torch
        n = 1 << 30
        buf = torch.ones(n, dtype=torch.int32, device=self.dev)
        out = torch.zeros(1, dtype=torch.int64, device=self.dev)
        s = torch.cuda.current_stream().cuda_stream or None
        best = 0.0
        for i in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._lib.check(self.lib.gm_stream_ceiling(buf.data_ptr(), n, out.data_ptr(), s), "gm_stream_ceiling")
            e1.record()
            torch.cuda.synchronize()
            if i:
                best = max(best, 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del buf
        return best


# ---------------------------------------------------------------------------------------------------------------
# HBM-side traffic: the same workloads under rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE)
# ---------------------------------------------------------------------------------------------------------------
def traffic_worker(a):
    """Child of measure_traffic(): runs every requested workload (1 warm-up + `steps` launches), prints the launch counts."""
    a.first_call_repeats = 0  # (the child's launches are counted: steps + 1 per workload, nothing else of the same kernels)
    r = Runner(a)
    todo = [w for w in (a.configs.split(",") if a.configs != "all" else ["tc", "diamond", "clique4", "motif3"])]
    graphs, launches = {}, {}
    dbl = [C.c_double(0) for _ in range(4)]
    for w in todo:
        scale, ef = (a.scale or WORKLOADS[w][0]), (a.ef or WORKLOADS[w][1])
        prefix = dataset_prefix(a, w)
        key = prefix or (scale, ef)
        if key not in graphs:
            for bg in graphs.values():
                bg.free()
            graphs.clear()
            graphs[key] = build_graph(a, r.local_rank, scale, ef, prefix)
        # a MARKER dispatch in front of every workload (the tiny issue-calibration kernel): two workloads may launch kernels of the same
        # name (TC and the 3-motif default both run tch_kernel), the parent attributes a row to the workload whose marker precedes it
        r._lib.check(r.lib.gm_issue_calib(0, 1, 1, *[C.byref(x) for x in dbl]), "marker")
        r.run(w, graphs[key], a.steps, 1)
        launches[w] = a.steps + 1
    print("TRAFFIC_WORKER " + json.dumps(launches), flush=True)
    r.close()


def dataset_prefix(a, workload):
    """real dataset for a BASELINE config when --data-dir holds it, else '' (synthetic stand-in)"""
    if a.graph:
        return a.graph
    if not a.data_dir:
        return ""
    name = {"tc": "livej", "diamond": "livej", "clique4": "com-orkut"}.get(workload)
    p = os.path.join(a.data_dir, name, "graph") if name else ""
    return p if p and os.path.exists(p + ".meta.txt") else ""


ISSUE_PASSES = (("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"), ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"))


def measure_traffic(a, workloads, share=(0, 1), device=0, issue=True):
    """{workload: {"fetch_bytes", "write_bytes", "launches", "kernels": {...}, "pmc": {kernel: {counter: per launch}}}} per launch, or (None, reason).
    share = (rank, world): the launches of the child cover that rank's share of the task chunks (N > 1: rank 0 measures ITS share on
    its own GPU while the other ranks wait; rank / world are launch arguments, no second GPU is involved).
    Passes (each its own rocprofv3 run, --kernel-trace --pmc only): FETCH_SIZE, WRITE_SIZE, and -- issue = True -- the two groups of ISSUE_PASSES
    behind the issue-side roofline; a failing issue pass costs only the `issue` object, never the traffic."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {w: {"kernels": {}, "pmc": {}} for w in workloads}
    tmp = tempfile.mkdtemp(prefix="gm_traffic_", dir="/tmp")
    try:
        for pi, group in enumerate((("FETCH_SIZE",), ("WRITE_SIZE",)) + (ISSUE_PASSES if issue else ())):
            counter = group[0]
            optional = pi >= 2
            d = os.path.join(tmp, f"pass{pi}")
            cmd = [exe, "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-worker", "--configs", ",".join(workloads), "--steps", "2",
                   "--traffic", "off", "--no-cpu-baseline", "--seed", str(a.seed)]
            for flag, val in (("--scale", a.scale), ("--ef", a.ef)):
                if val:
                    cmd += [flag, str(val)]
            if share[1] > 1:
                cmd += ["--share-rank", str(share[0]), "--share-world", str(share[1]), "--policy", str(a.policy)]
            for flag, val in (("--graph", a.graph), ("--data-dir", a.data_dir), ("--uniform", a.uniform), ("--powerlaw", a.powerlaw), ("--community", getattr(a, "community", "")), ("--tune", a.tune)):
                if val:
                    cmd += [flag, val]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GM_BENCH_FORCE_DIST", "GM_BENCH_ONE_GPU", "GM_BENCH_BACKEND", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
                env.pop(k, None)
            vis = [x for x in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if x]
            env["HIP_VISIBLE_DEVICES"] = vis[device] if device < len(vis) else str(device)  # the child sees this rank's GPU as device 0
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=env, timeout=900)
            m = re.search(r"TRAFFIC_WORKER (\{.*\})", r.stdout)
            if r.returncode != 0 or not m:
                if optional:
                    for w in workloads:
                        out[w]["pmc_error"] = f"rocprofv3 --pmc {' '.join(group)} failed (rc {r.returncode})"
                    continue
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            launches = json.loads(m.group(1))
            import csv

            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") in group:
                        rows.append((int(row.get("Dispatch_Id") or row.get("Correlation_Id") or 0), row.get("Kernel_Name", ""), row.get("Counter_Name"),
                                     float(row.get("Counter_Value", 0))))
            if not rows:
                if optional:
                    continue
                return None, f"no {counter} rows in the rocprofv3 output"
            rows.sort(key=lambda x: x[0])
            seg_sums, seg, in_marker = [], -1, False  # per workload (in the worker's order): (kernel name, counter) -> sum
            for did, k, cn, v in rows:
                if TRAFFIC_MARKER in k:
                    if not in_marker:  # (one boundary however many dispatches / counter rows the marker call makes)
                        seg, in_marker = seg + 1, True
                        seg_sums.append({})
                    continue
                in_marker = False
                if seg >= 0:
                    seg_sums[seg][(k, cn)] = seg_sums[seg].get((k, cn), 0.0) + v
            if len(seg_sums) != len(workloads):
                if optional:
                    continue
                return None, f"{len(seg_sums)} marker dispatches for {len(workloads)} workloads in the {counter} pass"
            for wi, w in enumerate(workloads):
                sums = seg_sums[wi]
                if optional:
                    for (kname, cn), v in sums.items():
                        if any(p in kname for p in KERNELS.get(w, [])):
                            out[w]["pmc"].setdefault(kname[:60], {})[cn] = v / launches[w]
                    continue
                tot = 0.0
                for (kname, cn), v in sums.items():
                    if any(p in kname for p in KERNELS.get(w, [])):
                        tot += v
                        out[w]["kernels"][kname[:60]] = out[w]["kernels"].get(kname[:60], {})
                        out[w]["kernels"][kname[:60]][counter] = v / launches[w]
                # counter unit: KB. gfx950: FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced stream (MI355X_MICROARCH.md,
                # re-checked for the dword-per-lane width these kernels use: profiles/r01/traffic_fetch_write_summary.txt) -> x2;
                # WRITE_SIZE is uncalibrated (x1)
                scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0
                if counter == "FETCH_SIZE" and tot <= 0:
                    return None, f"no FETCH_SIZE rows matched the kernels of {w} ({KERNELS.get(w)}): " + ", ".join(sorted(k[0][:40] for k in sums)[:6])
                out[w]["fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = tot * scale / launches[w]
                out[w]["launches"] = launches[w]
        return out, ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same workloads in this run; FETCH x2 (gfx950 calibration), KB units"
                     + (f"; share of rank {share[0]} of {share[1]} on one GPU" if share[1] > 1 else ""))
    except Exception as e:  # a report, never a reason to lose the bench line
        return None, f"traffic measurement failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def issue_roofline(pmc):
    """The ISSUE-side roofline of a workload's dominant kernel (most GRBM_GUI_ACTIVE cycles) out of the counters of ISSUE_PASSES, priced with the
    calibrated issue costs of profiles/r03/issue_calibration.txt as scripts/summarize_pmc.py does: a wave64 VALU instruction occupies its
    SIMD 2.4 (add / logic / shift / mov) .. 4.1 cycles (the rest) -- SQ_INSTS_VALU does not tell the classes apart, hence a range; a SALU
    instruction the CU's one scalar unit 1.0 cycle; SQ_LDS_IDX_ACTIVE = busy cycles of the LDS.  `binds`: the busiest unit, or the waves' own
    latency when no unit is above 0.6 and the waves wait more than half of their cycles."""
    best = None
    for k, c in (pmc or {}).items():
        if c.get("GRBM_GUI_ACTIVE", 0) > (best[1].get("GRBM_GUI_ACTIVE", 0) if best else 0):
            best = (k, c)
    if not best:
        return None
    k, c = best
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # (summed over the 8 XCDs)
    v, sa, lds = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_SALU", 0.0), c.get("SQ_LDS_IDX_ACTIVE")
    r = {"kernel": k[:48], "valu": [round(v * 2.4 / 1024 / cyc, 3), round(min(v * 4.1 / 1024 / cyc, 9.99), 3)], "salu": round(sa / 256 / cyc, 3)}
    if lds is not None:
        r["lds"] = round(lds / 256 / cyc, 3)
        r["lds_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(lds, 1.0), 3)
    if c.get("SQ_WAVE_CYCLES"):
        r["waiting"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    units = {"valu": r["valu"][1], "salu": r["salu"], "lds": r.get("lds", 0.0)}
    top = max(units, key=units.get)
    r["binds"] = top if units[top] >= 0.6 or r.get("waiting", 0.0) <= 0.5 else "latency of the waves' own chains (no unit above 0.6 busy)"
    r["kernel_cycles"] = int(cyc)
    return r


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): the reference's binaries (oracle/_ref) and / or the oracle, on this box's host cores
# ---------------------------------------------------------------------------------------------------------------
def median(xs):
    s = sorted(xs)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def run_reference(exe_name, args, pattern_time, pattern_count, threads, runs, timeout=600):
    exe = os.path.join(ROOT, "oracle", "_ref", exe_name)
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread")
    times, count = [], None
    for _ in range(runs):
        r = subprocess.run([exe, *args], capture_output=True, text=True, env=env, timeout=timeout)
        m = re.search(pattern_time, r.stdout)
        c = re.findall(pattern_count, r.stdout)
        if r.returncode != 0 or not m or not c:
            return None
        times.append(float(m.group(1)))
        count = [int(x) for x in c]
    return {"seconds": median(times), "all_seconds": [round(t, 3) for t in times], "count": count}


# reduced scale (same generator, same seed and edge factor) on which the reference binary of a slow workload finishes in
# 10-40 s on the host: its whole-graph run there is the bounded CPU sample of that workload
REDUCED_SCALE = {"clique4": 20, "motif3": 22, "motif3e": 21}  # (motif3: motif_omp_formula, the solver gm_motif takes; motif3e: motif_omp_base, the enumeration)


def cpu_baselines(a, r, recs, graphs):
    """cpu_baseline object per workload, all from the REFERENCE's own binaries (oracle/_ref, their own Timer line = the
    solver loop only, src/triangle/omp_base.cc:12-24), OMP_NUM_THREADS = the oracle's thread count, OMP_PROC_BIND=spread:
      tc       tc_omp_base on the whole graph, median of 3 runs (+ the oracle restatement as `port`);
      diamond  sgl_omp_base on the whole graph, one run (tens of seconds);
      clique4 / motif3   minutes at full size (recorded: tests/golden/fullsize.json), so the bounded sample is the WHOLE graph of
               the same generator at a reduced scale (REDUCED_SCALE), one run; the GPU runs that graph too and the counts
               are compared. A vertex-strided sample of the full graph was tried and rejected: the loop is vertex-parallel, so
               a sample that contains a hub row measures that row's serial time, not the machine's throughput."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O  # the CPU oracle, timed as the OpenMP baseline ("port"): never the product path

    threads = O.num_threads()
    out = {}
    tmpdirs = []

    def save(bg):
        tmp = tempfile.mkdtemp(prefix="gm_ref_", dir="/tmp")
        tmpdirs.append(tmp)
        h = bg.sym.download()
        h.save(os.path.join(tmp, "graph"))
        return os.path.join(tmp, "graph"), h

    def ref_record(exe, args, tasks, count_pat, gpu_count, runs, what, nvals=1):
        ref = run_reference(exe, args, r"runtime(?: \[[a-z_]+\])? = ([0-9.eE+-]+)", count_pat, threads, runs, timeout=900)
        if not ref:
            return None
        cnt = ref["count"][-nvals:] if nvals > 1 else ref["count"][-1]
        return {"value": round(tasks / ref["seconds"] / 1e6, 3), "unit": "Medges/s", "cores": threads, "kind": "reference",
                "seconds": round(ref["seconds"], 3), "runs_seconds": ref["all_seconds"], "count_matches_gpu": bool(cnt == gpu_count),
                "cpu_count": cnt, "sample": what, "host_cpus": os.cpu_count()}

    try:
        prefixes = {}
        for rec in recs:
            w, bg = rec["workload"], graphs[rec["workload"]]
            progress(f"  cpu baseline of {w}")
            try:
                if w in ("tc", "diamond"):
                    if bg.name not in prefixes:
                        prefixes[bg.name] = save(bg)
                    prefix, h = prefixes[bg.name]
                if w == "tc":
                    odag = O.orient(O.OGraph(h.row_ptr, h.col_idx))
                    cal = 256
                    t = time.perf_counter()
                    O.tc_sample(odag, cal, 1)
                    stride = max(1, int((time.perf_counter() - t) * cal / max(a.cpu_seconds, 1e-3)))
                    t = time.perf_counter()
                    cnt, tasks = O.tc_sample(odag, stride, 0)
                    dt = time.perf_counter() - t
                    port = {"value": round(tasks / dt / 1e6, 3), "unit": "Medges/s", "cores": threads, "kind": "port", "seconds": round(dt, 2),
                            "sample": f"oracle gmo_tc_sample: vertices u = 0 mod {stride} of the same DAG ({tasks} of {rec['tasks']} task "
                                      f"edges), OpenMP schedule(dynamic,1)", "host_cpus": os.cpu_count()}
                    if stride == 1:
                        port["count_matches_gpu"] = bool(cnt == rec["count"])
                    ref = None if a.no_ref_baseline else ref_record(
                        "tc_omp_base", [prefix], rec["tasks"], r"total_num_triangles = (\d+)", rec["count"], 3,
                        "tc_omp_base (reference binary, g++ -O3 -fopenmp, its own orientation + Timer) on the whole graph, median of 3 runs")
                    if ref:
                        ref["port"] = {k: port[k] for k in ("value", "seconds", "sample")}
                    out[w] = ref or port
                elif w == "diamond" and not a.no_ref_baseline:
                    out[w] = ref_record("sgl_omp_base", [prefix, "diamond"], rec["tasks"], r"total_num = (\d+)", rec["count"], 1,
                                        "sgl_omp_base diamond (reference binary, its own Timer) on the whole graph, ONE run (tens of "
                                        "seconds; the reference's methodology is the mean of 3)")
                elif w in ("clique4", "motif3") and not a.no_ref_baseline and bg.name.startswith("rmat"):
                    # (motif3: like with like -- the formula solver gm_motif takes against motif_omp_formula, and the enumeration kernels
                    # of `per_edge_variant` against motif_omp_base, each on its own reduced graph)
                    for wk in ([w] if w == "clique4" else ["motif3", "motif3e"]):
                        scale = min(REDUCED_SCALE[wk], a.scale or WORKLOADS[w][0])
                        ef = a.ef or WORKLOADS[w][1]
                        small = build_graph(a, r.local_rank, scale, ef)
                        try:
                            gpu = r.run(wk, small, 1, 1, solo=True)  # the HIP path on the reduced graph: count for the comparison
                            sp, _h = save(small)
                            del _h
                            exe, args = {"clique4": ("clique_omp_base", [sp, "4"]), "motif3": ("motif_omp_formula", [sp, "3"]),
                                         "motif3e": ("motif_omp_base", [sp, "3"])}[wk]
                            if wk == "clique4":
                                rr = ref_record(exe, args, gpu["tasks"], r"num_4-cliques = (\d+)", gpu["count"], 1, "")
                            else:
                                # (motif_omp_formula takes ~5 s on its reduced graph: three runs, the median, like the TC baseline)
                                rr = ref_record(exe, args, gpu["tasks"], r"pattern \d+: (\d+)", gpu["count"], 3 if wk == "motif3" else 1, "", nvals=2)
                            if rr:
                                rr["sample"] = (f"{exe} {args[1]} (reference binary, its own Timer) on the WHOLE graph of the same generator at reduced "
                                                f"scale: {small.name} ({gpu['tasks']} task edges instead of {rec['tasks']}), {'median of 3 runs' if wk == 'motif3' else 'ONE run'}; GPU "
                                                f"({'formula solver' if wk == 'motif3' else 'enumeration kernels' if wk == 'motif3e' else 'default path'}) "
                                                f"on that graph: {gpu['kernel_ms_avg']:.3f} ms")
                                rr["gpu_ms_on_sample_graph"] = round(gpu["kernel_ms_avg"], 4)
                                rr["count_matches_gpu_on"] = small.name
                                out[wk] = rr
                        finally:
                            small.free()
            except Exception as e:
                print(f"[bench] cpu baseline of {w} skipped: {e}", file=sys.stderr)
    finally:
        for tmp in tmpdirs:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def config1_record(a):
    """BASELINE configs[0]: `tc_omp_base inputs/citeseer/graph` -- the CPU plumbing case (loader + orientation + OMP solver),
    run as the reference binary and as the oracle's CLI; plus the same graph through the HIP path."""
    prefix = os.path.join(ROOT, "tests", "fixtures", "citeseer", "graph")
    rec = {"id": 1, "config": "tc_omp_base on inputs/citeseer/graph (triangle count, CPU-only plumbing)", "graph": "citeseer",
           "expected": 1166}
    for kind, exe in (("reference", os.path.join(ROOT, "oracle", "_ref", "tc_omp_base")), ("port", os.path.join(ROOT, "oracle", "bin", "tc_omp_base"))):
        if os.path.exists(exe):
            r = subprocess.run([exe, prefix], capture_output=True, text=True, timeout=120)
            m = re.search(r"total_num_triangles = (\d+)", r.stdout)
            rec[f"{kind}_count"] = int(m.group(1)) if m else None
    try:
        from graphminer_amd import Graph, TCSolver

        with Graph(prefix).to_device(int(os.environ.get("LOCAL_RANK", "0"))) as sym:
            dag = sym.orient()
            rec["gpu_count"] = TCSolver(dag)
            dag.free()
    except Exception as e:
        rec["gpu_count"] = f"error: {e}"
    # a CPU count must actually have been produced in THIS run (the binaries under oracle/_ref and oracle/bin are built by
    # __graft_entry__.build(), not tracked): a missing one is never counted as a match
    cpu = {k: rec[k] for k in ("reference_count", "port_count") if rec.get(k) is not None}
    if cpu:
        rec["count_matches_cpu"] = bool(all(c == 1166 for c in cpu.values()) and rec.get("gpu_count") == 1166)
        rec["cpu_count_source"] = "this run: " + " + ".join(sorted(cpu))
    else:
        rec["count_matches_cpu"] = None
        rec["cpu_count_source"] = "no CPU binary produced a count in this run (oracle/_ref and oracle/bin not built)"
    return rec


# ---------------------------------------------------------------------------------------------------------------
def finish_record(rec, a, world, ab, floor, traffic, traffic_src, cpu, known, stream_gbs, own=None):
    """the JSON sub-record of one workload"""
    t = rec["kernel_ms_avg"] * 1e-3
    step_t = rec["elapsed"] / rec["steps"]
    out = {
        "workload": rec["workload"], "graph": rec["graph"], "nv": rec["nv"], "ne_sym": rec["ne_sym"], "tasks": rec["tasks"],
        "max_degree": rec["max_degree"], "steps": rec["steps"], "ms_per_step": round(rec["ms_per_step"], 4),
        "kernel_ms_avg": round(rec["kernel_ms_avg"], 4), "value": round(rec["tasks"] / step_t / 1e6, 3), "unit": "Medges/s",
        "count": rec["count"], "matches_per_sec": round((rec["count"][1] if isinstance(rec["count"], list) else rec["count"]) / step_t, 1),
        "first_call_ms": round(rec["first_call_ms"], 2), "first_call_runs_ms": rec.get("first_call_runs_ms"), "setup_ms": rec["setup_ms"], "end_to_end_ms": round(rec.get("end_to_end_ms", rec["first_call_ms"]), 2),
        "single_shot_ms": round(rec["single_shot_ms"], 2) if rec.get("single_shot_ms") is not None else None,
        "per_gpu_kernel_ms": {"max": round(max(rec["per_gpu_kernel_ms"]), 4), "mean": round(sum(rec["per_gpu_kernel_ms"]) / len(rec["per_gpu_kernel_ms"]), 4),
                              "all": [round(x, 4) for x in rec["per_gpu_kernel_ms"]],
                              "skew_max_over_mean": round(max(rec["per_gpu_kernel_ms"]) / max(sum(rec["per_gpu_kernel_ms"]) / len(rec["per_gpu_kernel_ms"]), 1e-9), 4)},
        "grid": rec["stats"],
    }
    roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": " + ".join(k.rstrip(",") + (">" if k.endswith(",") else "") for k in KERNELS.get(rec["workload"], ["?"])),
            "compulsory_floor_bytes": floor, "stream_ceiling_GBs": round(stream_gbs, 1) if stream_gbs else None}
    alg_gbs = None
    if ab is not None and t > 0:
        per_launch = ab / world  # each rank's kernel covers ~1/world of the task chunks
        alg_gbs = per_launch / t / 1e9
        roof.update({"algorithmic_bytes_per_launch": int(per_launch), "algorithmic_GBs": round(alg_gbs, 2),
                     "algorithmic_frac": round(alg_gbs / HBM_PEAK_GBS, 5)})
    # A launch with a hub-corner kernel on the matrix cores (gm_ctc.hip).  `achieved` / `frac` stay WHOLE-LAUNCH figures (every
    # kernel's counter traffic, the corner's included, over kernel_ms_avg -- the time the library's two events bracket); the streamed
    # kernels alone (their traffic over launch - corner) go under `streamed_kernels`, the corner kernel under `corner_kernel` against the
    # dense FP4 MFMA peak.
    corner_ms = float(rec.get("corner_ms_avg") or 0.0)
    if corner_ms > 0 and corner_ms < rec["kernel_ms_avg"]:
        t_str = (rec["kernel_ms_avg"] - corner_ms) * 1e-3
        ctr = {"ms": round(corner_ms, 4), "bound": "mfma", "peak": 10000.0, "unit": "TFLOP/s (dense FP4, v_mfma_scale_f32_32x32x64_f8f6f4)"}
        if own is not None and own.get("parts", {}).get("corner_operand_chunks"):
            mf = own["parts"]["corner_operand_chunks"] // 32768 * 512  # MFMA instructions: 8 waves x 64 per (block pair, chunk)
            ctr.update({"mfma_instructions": int(mf), "achieved": round(mf * 131072 / (corner_ms * 1e-3) / 1e12, 1)})
            ctr["frac"] = round(ctr["achieved"] / ctr["peak"], 4)
        streamed = {"ms": round(t_str * 1e3, 4)}
        if traffic and traffic.get("kernels"):
            cb = sum(v.get("FETCH_SIZE", 0) * 2048.0 + v.get("WRITE_SIZE", 0) * 1024.0 for k, v in traffic["kernels"].items() if CORNER_KERNEL in k)
            ctr["traffic"] = int(cb)
            sb = sum(v.get("FETCH_SIZE", 0) * 2048.0 + v.get("WRITE_SIZE", 0) * 1024.0 for k, v in traffic["kernels"].items() if CORNER_KERNEL not in k)
            streamed.update({"traffic": int(sb), "GBs": round(sb / t_str / 1e9, 2), "frac": round(sb / t_str / 1e9 / HBM_PEAK_GBS, 5)})
        if own is not None and own.get("parts", {}).get("corner_operand_chunks"):
            ob = (own["bytes"] - own["parts"]["corner_operand_chunks"]) / world
            streamed.update({"own_bytes": int(ob), "own_frac": round(ob / t_str / 1e9 / HBM_PEAK_GBS, 5)})
        roof["corner_kernel"] = ctr
        roof["streamed_kernels"] = streamed
        roof["streamed_kernels_ms"] = streamed["ms"]
    tr_gbs = None
    if traffic:
        tb = traffic.get("fetch_bytes", 0.0) + traffic.get("write_bytes", 0.0)
        tr_gbs = tb / t / 1e9 if t > 0 else None
        roof.update({"traffic": int(tb), "traffic_fetch_bytes": int(traffic.get("fetch_bytes", 0)), "traffic_write_bytes": int(traffic.get("write_bytes", 0)),
                     "traffic_GBs": round(tr_gbs, 2) if tr_gbs else None, "traffic_frac": round(tr_gbs / HBM_PEAK_GBS, 5) if tr_gbs else None,
                     "traffic_source": traffic_src, "traffic_kernels": traffic.get("kernels")})
    else:
        roof.update({"traffic": None, "traffic_source": traffic_src})
    # `own_*`: the bytes this library's kernels must move by construction (own_bytes_device), recomputable from the graph
    own_gbs = None
    if own is not None and t > 0:
        per_launch_own = own["bytes"] / world
        own_gbs = per_launch_own / t / 1e9
        roof.update({"own_bytes_per_launch": int(per_launch_own), "own_streamed_keys_per_launch": int(own["streamed_keys"] / world),
                     "own_parts_whole_graph": own["parts"], "own_GBs": round(own_gbs, 2), "own_frac": round(own_gbs / HBM_PEAK_GBS, 5),
                     "keys_per_second": round(own["streamed_keys"] / world / t, 1)})
    # `achieved` / `frac`: the COUNTER traffic of the launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate passes)
    # / kernel time / 8 TB/s whenever the PMC pass succeeded -- what SURVEY 8(d) calls achieved-BW = measured bytes / t.  `own_*` (what the
    # kernels must move by construction) and `algorithmic_*` (the reference's loop nest) stay beside it; without counters
    # the own figure is used and `frac_basis` says so.
    if tr_gbs is not None:
        roof.update({"achieved": round(tr_gbs, 2), "frac": round(tr_gbs / HBM_PEAK_GBS, 5),
                     "frac_basis": "counter traffic (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE, separate passes) / HIP-event kernel time / 8 TB/s"})
    elif own_gbs is not None:
        roof.update({"achieved": round(own_gbs, 2), "frac": round(own_gbs / HBM_PEAK_GBS, 5),
                     "frac_basis": "NO counter pass in this run: own-algorithm bytes (DESIGN.md: streamed keys + descriptors + rows staged once + offsets + arena) / kernel time / 8 TB/s"})
    elif alg_gbs is not None and alg_gbs <= HBM_PEAK_GBS:
        roof.update({"achieved": round(alg_gbs, 2), "frac": round(alg_gbs / HBM_PEAK_GBS, 5), "frac_basis": "algorithmic bytes (SURVEY 8d) / kernel time / 8 TB/s"})
    else:
        roof.update({"achieved": round(alg_gbs, 2) if alg_gbs else None, "frac": None,
                     "frac_basis": "no counter traffic available and the algorithmic formula exceeds the peak: not a roofline"})
    if alg_gbs is not None:
        roof["frac_8d_valid"] = bool(alg_gbs <= HBM_PEAK_GBS)
        if alg_gbs > HBM_PEAK_GBS:
            roof["frac_8d_note"] = ("SURVEY 8(d) charges both lists of every edge (the reference's loop nest); this kernel keeps the longer row in LDS and streams "
                                    "only the shorter list, so the 8(d) bytes / time exceed the HBM peak and are not a bound for it")
    if stream_gbs and roof.get("achieved"):
        roof["frac_of_stream_ceiling"] = round(roof["achieved"] / stream_gbs, 5)
    if rec["workload"] in ("tc", "motif3", "motif3f"):
        roof["note"] = ("tch_kernel streams the SHORTER list of every DAG edge against the longer one kept as a hashed set in LDS (sum min(d+(u), d+(v)) "
                        "keys; the section-8(d) formula charges N+(u) and N+(v) per edge)")
    elif rec["workload"] == "diamond" and (world <= 1 or rec.get("diamond_supports_across_ranks")):
        roof["note"] = ("one GPU: |N(u) ^ N(v)| of every edge = its triangles, counted in ONE pass over the triangles of the DAG (sup_kernel: the triangle "
                        "kernel with three increments per match), then sum C(t, 2); the section-8(d) formula charges one intersection of the symmetric "
                        "lists per edge. A device-scope atomic moves a 64-byte fabric transaction for its 4 bytes: counter traffic is well above the own bytes")
    elif rec["workload"] in ("diamond", "motif3e"):
        roof["note"] = "rows > 1024 entries are hashed sets in LDS (gm_hrow.hip): every partner list is fetched about once"
    if rec["nv"] * 8 + rec["ne_sym"] * 4 <= (256 << 20):
        roof["mall_note"] = ("the CSR fits the 256 MiB Infinity Cache: FETCH_SIZE counts MALL hits too (MI355X_MICROARCH.md), so `traffic` here is "
                             "fabric traffic, most of it served from the Infinity Cache, not HBM reads")

    if traffic and traffic.get("pmc"):
        iss = issue_roofline(traffic["pmc"])
        if iss:
            roof["issue"] = iss
    out["roofline"] = roof
    if cpu:
        fs = full_size_cpu(rec["workload"], rec["graph"], rec["tasks"])
        if fs:  # the stated baseline ON the bench graph, beside this run's reduced-scale check
            fs["count_equals_gpu_count_of_this_run"] = bool(known is not None and known["count"] == rec["count"])
            cpu = dict(cpu, full_size=fs)
        out["cpu_baseline"] = cpu
    # count check: against the CPU count of this run when one exists, else against the recorded full-size oracle answers
    chk = {}
    if cpu and "count_matches_gpu" in cpu and ("count_matches_gpu_on" not in cpu or cpu["count_matches_gpu_on"] == rec["graph"]):  # (a "reduced" graph that IS the bench graph: small --scale runs)
        chk = {"count_matches_cpu": cpu["count_matches_gpu"], "cpu_count_source": f"this run: {cpu['kind']} ({cpu['sample'][:60]}...)"}
    elif known is not None and cpu and "count_matches_gpu_on" in cpu:  # full size: recorded answer; reduced scale: this run
        chk = {"count_matches_cpu": bool(known["count"] == rec["count"]) and bool(cpu["count_matches_gpu"]),
               "cpu_count_source": known["source"] + f"; and this run's reference binary on {cpu['count_matches_gpu_on']}"}
    elif known is not None:
        chk = {"count_matches_cpu": bool(known["count"] == rec["count"]), "cpu_count_source": known["source"]}
    else:
        chk = {"count_matches_cpu": None, "cpu_count_source": "no full CPU count for this graph (sampled baseline only)"}
    out.update(chk)
    return out


def full_size_cpu(workload, graph_name, tasks):
    """the recorded run of the reference's binary on the WHOLE bench graph (minutes: not repeated inside bench.py) -- tests/golden/fullsize.json"""
    try:
        e = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json"))).get(f"{workload}:{graph_name}") or {}
        if "reference_seconds" not in e:
            return None
        return {"seconds": e["reference_seconds"], "threads": e.get("threads"), "binary": e.get("binary"), "recorded": e.get("recorded"),
                "value": round(tasks / e["reference_seconds"] / 1e6, 4), "unit": "Medges/s", "count_equals_gpu_count_of_this_run": None}
    except Exception:
        return None


def known_answer(a, workload, graph_name):
    """recorded CPU answers: README values for the real datasets, full-size oracle runs for the default stand-ins"""
    try:
        if graph_name.startswith("file:"):
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["_readme_known_answers"]
            for ds, e in gold.items():
                if f"/{ds}/" in graph_name and workload in e:
                    return {"count": e[workload], "source": f"README known answer of {ds} (tests/golden/golden.json::_readme_known_answers)"}
            return None
        full = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))
        e = full.get(f"{workload}:{graph_name}")
        return {"count": e["count"], "source": e["source"]} if e else None
    except Exception:
        return None


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here (the reference's *_multigpu binaries take
    n_gpu as an argument and spawn their own threads, src/clique/multigpu.cu:20,109-115) -- re-exec under torch.distributed.run, one
    process per GPU, rendezvous on 127.0.0.1. Under an external launcher (WORLD_SIZE set) nothing happens here."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ or a.traffic_worker:
        return
    import socket

    import torch

    have = torch.cuda.device_count()
    if os.environ.get("GM_BENCH_ONE_GPU") != "1" and have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} HIP device(s) visible on this node (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '')!r}); "
                         f"run with --gpus {max(have, 1)}, or GM_BENCH_ONE_GPU=1 for the one-GPU dry run of the {a.gpus}-rank path")
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GM_BENCH_SELF_LAUNCHED="1")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


SHORT_BASIS = (("counter traffic", "counter_traffic: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE / HIP-event kernel time / 8 TB/s"),
               ("NO counter pass", "own_bytes (no counter pass in this run) / kernel time / 8 TB/s"),
               ("algorithmic bytes", "algorithmic bytes (SURVEY 8d) / kernel time / 8 TB/s"))


def compact_roofline(r):
    """the roofline object of the stdout line: the contract's keys + the three byte figures, no prose"""
    keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "own_bytes_per_launch", "own_frac", "algorithmic_bytes_per_launch",
            "algorithmic_GBs", "frac_8d_valid", "compulsory_floor_bytes", "stream_ceiling_GBs", "frac_of_stream_ceiling", "streamed_kernels_ms", "whole_launch_frac")
    out = {k: r[k] for k in keys if k in r}
    if "kernel" in out:
        out["kernel"] = out["kernel"][:96]
    if r.get("corner_kernel"):
        out["corner_kernel"] = {k: r["corner_kernel"][k] for k in ("ms", "bound", "achieved", "peak", "frac", "traffic") if k in r["corner_kernel"]}
    if isinstance(r.get("issue"), dict):
        out["issue"] = {k: (r["issue"][k][:32] if k == "kernel" else r["issue"][k]) for k in ("kernel", "valu", "salu", "lds", "lds_conflict", "waiting", "binds") if k in r["issue"]}
    if r.get("streamed_kernels"):  # (round 6: `frac` is the whole launch; the streamed kernels alone beside it)
        out["streamed_kernels"] = {k: r["streamed_kernels"][k] for k in ("ms", "traffic", "frac", "own_frac") if k in r["streamed_kernels"]}
    basis = r.get("frac_basis") or ""
    out["frac_basis"] = next((short for head, short in SHORT_BASIS if basis.startswith(head)), basis[:80])
    return out


def compact_cpu(c):
    if not c:
        return None
    out = {k: c[k] for k in ("value", "unit", "cores", "kind", "seconds", "count_matches_gpu", "count_matches_gpu_on", "full_size", "carried_from") if k in c}
    out["sample"] = c.get("sample", "")[:160]
    return out


def compact_line(out, detail_path):
    """The ONE stdout line (< 4 KB, the driver keeps an 8 KB tail): the contract's keys with the headline's roofline and
    cpu_baseline, a six-field summary per BASELINE config, and the path of the full record. (The reference prints one short TEPS line,
    src/triangle/gpu_base.cu:68-71.)"""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data") if k in out}
    line["config"] = {k: out["config"][k] for k in ("workload", "graph", "nv", "ne_sym", "tasks", "parallelism") if k in out["config"]}
    line.update({"count": out["count"], "count_matches_cpu": out.get("count_matches_cpu"), "kernel_ms_avg": out["kernel_ms_avg"],
                 "first_call_ms": out["first_call_ms"], "first_call_runs_ms": out.get("first_call_runs_ms"), "rccl_ranks_seen": out.get("rccl_ranks_seen", 1)})
    line["per_gpu_kernel_ms"] = {k: out["per_gpu_kernel_ms"][k] for k in ("max", "mean")}
    line["roofline"] = compact_roofline(out["roofline"])
    line["cpu_baseline"] = compact_cpu(out.get("cpu_baseline"))
    if "configs" in out:
        cfgs = []
        for c in out["configs"]:
            if c["id"] == 1:
                cfgs.append({"id": 1, "workload": "tc_omp_base citeseer", "count": c.get("gpu_count"), "count_ok": c.get("count_matches_cpu")})
                continue
            rf, cb = c["roofline"], c.get("cpu_baseline") or {}
            e = {"id": c["id"], "workload": c["workload"], "graph": c["graph"], "kernel_ms": c["kernel_ms_avg"], "value": c["value"], "count": c["count"],
                 "count_ok": c.get("count_matches_cpu"), "frac": rf.get("frac"), "traffic_GB": round(rf["traffic"] / 1e9, 2) if rf.get("traffic") else None,
                 "own_frac": rf.get("own_frac"), "first_call_ms": c["first_call_ms"], "single_shot_ms": c.get("single_shot_ms"), "cpu_value": cb.get("value"), "cpu_kind": cb.get("kind")}
            if isinstance(rf.get("issue"), dict):  # which unit binds the dominant kernel and how busy it is (issue-side roofline; the detail file has all of it)
                iss = rf["issue"]
                busy = {"valu": iss["valu"][1], "salu": iss["salu"], "lds": iss.get("lds", 0.0)}
                e["binds"] = f"{iss['binds'][:7]} {max(busy.values()):.2f}, waiting {iss.get('waiting', 0):.2f}"
            if cb.get("full_size"):
                e["cpu_full_size_s"] = cb["full_size"].get("seconds")
            if "per_edge_variant" in c and "kernel_ms_avg" in c["per_edge_variant"]:
                e["enumeration_kernel_ms"] = c["per_edge_variant"]["kernel_ms_avg"]
                e["enumeration_frac"] = c["per_edge_variant"].get("frac")
                try:  # both CPU pairs of config 5: motif_omp_formula <-> the formula ms above, motif_omp_base <-> the enumeration ms
                    fs = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json"))).get(f"motif3:{c['graph']}") or {}
                    if fs.get("enumeration_reference_seconds"):
                        e["enumeration_cpu_full_size_s"] = fs["enumeration_reference_seconds"]
                except Exception:
                    pass
            cfgs.append(e)
        line["configs"] = cfgs
        line["all_counts_match_cpu"] = out.get("all_counts_match_cpu")
    if out.get("livejournal_standins"):  # LiveJournal-sized graphs of the two other shapes (ids 2a / 2b: TC, 3a / 3b: diamond)
        tags = {("tc", "uniform"): "2a", ("tc", "powerlaw"): "2b", ("tc", "community"): "2c", ("diamond", "uniform"): "3a", ("diamond", "powerlaw"): "3b",
                ("diamond", "community"): "3c"}  # (2c / 3c: LiveJournal's triangle density; Medges/s = 43 / kernel_ms: left out for the 4 KB)
        line["standins"] = [{"id": tags.get((x["workload"], x["graph"].split("_")[0]), "?"), "kernel_ms": x["kernel_ms_avg"], "count": x["count"],
                             **({"traffic_frac": x["traffic_frac"]} if x.get("traffic_frac") is not None else {"own_frac": x.get("frac")})} for x in out["livejournal_standins"]]
    if isinstance(out.get("tc_rmat24"), dict) and "roofline" in out["tc_rmat24"]:
        t = out["tc_rmat24"]
        line["tc_rmat24"] = {"kernel_ms": t["kernel_ms_avg"], "value": t["value"], "frac": t["roofline"].get("frac"), "own_frac": t["roofline"].get("own_frac")}
    line["detail"] = detail_path
    line["bench_wall_s"] = out.get("bench_wall_s")
    return line


def main():
    a = parse()
    if a.traffic_worker:
        return traffic_worker(a)
    self_launch(a)
    r = Runner(a)
    world, rank = r.world, r.rank
    single = bool(a.workload or a.graph or a.uniform or a.powerlaw or a.community)
    if single:
        todo = [(0, a.workload or "tc", WORKLOADS[a.workload or "tc"][3], None)]
    else:
        sel = ["tc", "diamond", "clique4", "motif3"] if a.configs == "all" else a.configs.split(",")
        todo = [c for c in BASELINE_CONFIGS if c[1] in sel]
        if not any(c[1] == "tc" for c in todo):
            todo.insert(0, BASELINE_CONFIGS[0])  # the headline is always measured

    t_start = time.perf_counter()
    graphs, recs, bytes_of, own_of = {}, [], {}, {}
    keep = {}
    for cid, w, desc, _ds in todo:
        scale, ef = (a.scale or WORKLOADS[w][0]), (a.ef or WORKLOADS[w][1])
        prefix = dataset_prefix(a, w)
        key = prefix or (scale, ef)
        if key not in keep:
            keep[key] = build_graph(a, r.local_rank, scale, ef, prefix)
        bg = keep[key]
        graphs[w] = bg
        progress(f"config {cid} {w} on {bg.name}: {a.warmup} + {a.steps} steps")
        rec = r.run(w, bg, a.steps, a.warmup)
        rec.update({"id": cid, "config": desc, "graph": bg.name, "input_build_s": bg.build_s})
        if rank == 0:
            bytes_of[w] = alg_bytes_device(w, bg, r.lib, rec["g"])
            # (diamond across ranks through the shared triangle pass: the one-GPU byte model, a rank's share of it)
            own_of[w] = own_bytes_device(w, bg, 1 if rec.get("diamond_supports_across_ranks") else r.world)
        recs.append(rec)

    out = None
    if rank == 0:
        progress("stream ceiling, counter passes")
        stream_gbs = r.stream_ceiling()
        # ---- HBM-side traffic ------------------------------------------------------------------------------------
        traffic, traffic_src = {}, "off"
        if a.traffic == "auto":
            # N > 1: ONE rocprofv3 pass pair of rank 0's share on rank 0's GPU (the other ranks wait in the closing barrier)
            # (the issue-side passes only at N = 1: a property of the kernels, and at N > 1 the other ranks wait in the closing barrier meanwhile)
            got, traffic_src = measure_traffic(a, [x["workload"] for x in recs], share=(0, world) if world > 1 else (a.share_rank, max(a.share_world, 1)),
                                               device=r.local_rank, issue=(world == 1))
            traffic = got or {}
        if (a.traffic == "file" or (a.traffic == "auto" and not traffic)) and os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")):
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            for x in recs:
                v = tj.get(f"{x['workload']}:{x['graph']}:n{world}")
                if v:
                    traffic[x["workload"]] = {"fetch_bytes": float(v), "write_bytes": 0.0}
            traffic_src = (traffic_src + "; fallback: " if a.traffic == "auto" else "") + "profiles/traffic.json (static, recorded " + str(tj.get("_recorded", "r01")) + ")"
        # ---- CPU baselines ---------------------------------------------------------------------------------------
        # (every N carries it: measured in this run at N = 1; at N > 1 carried from the N = 1 run of the same box and graphs when
        # that left its record in /tmp, else measured now by rank 0 while the other ranks wait)
        cpu = {}
        if not a.no_cpu_baseline:
            cpu_recs = [x for x in recs if x["workload"] in ("tc", "diamond", "clique4", "motif3")]
            cache = os.path.join(tempfile.gettempdir(), "gm_bench_cpu_" + re.sub(r"[^A-Za-z0-9_.-]", "_", "+".join(f"{x['workload']}.{x['graph']}" for x in cpu_recs))[:180] + ".json")
            if world > 1 and os.path.exists(cache) and time.time() - os.path.getmtime(cache) < 12 * 3600:
                try:
                    cpu = json.load(open(cache))
                    for v in cpu.values():
                        v["carried_from"] = "the N = 1 run of this box (same graphs, same binaries), " + time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(os.path.getmtime(cache)))
                except Exception:
                    cpu = {}
            if not cpu:
                progress("CPU baselines (reference binaries on the host cores)")
                cpu = cpu_baselines(a, r, cpu_recs, graphs)
                if world == 1 and not r.use_dist and cpu:
                    try:
                        json.dump(cpu, open(cache, "w"))
                    except Exception:
                        pass
        progress("records")
        subs = []
        for x in recs:
            ab, floor = bytes_of[x["workload"]]
            sub = finish_record(x, a, world, ab, floor, traffic.get(x["workload"]), traffic_src, cpu.get(x["workload"]),
                                known_answer(a, x["workload"], x["graph"]), stream_gbs, own_of.get(x["workload"]))
            sub = {"id": x["id"], "config": x["config"], **sub, "input_build_s": round(x["input_build_s"], 2)}
            if x["workload"] == "motif3":
                sub["solver"] = ("gm_motif default = the reference's formula solver (src/motif/omp_formula.cc:39-46): triangles of the oriented graph with the TC kernel, "
                                 "wedges = sum C(d,2) - 3T; the enumeration form (automine_3motif, tune[6] & 0x10000000) is measured in per_edge_variant")
            elif x["workload"] == "diamond":
                sub["solver"] = ("one GPU: |N(v0) ^ N(v1)| of every edge from one pass over the triangles of the oriented graph (edge supports, gm_sup.hip), then sum C(t,2); "
                                 "several ranks: one intersection of the two symmetric lists per edge (tune[6] & 0x10000000 selects it on one GPU)") if world <= 1 else \
                                ("several ranks: the same shared triangle pass, a rank's share each, ONE reduce-scatter of the support arrays (uint32 per DAG entry) over "
                                 "xGMI, sum C(t,2) of the rank's slice, then the all-reduce of the count" if x.get("diamond_supports_across_ranks") else
                                 "several ranks: one intersection of the two symmetric lists per edge (gm_hrow.hip, gm_chunk.h)")
            if x["workload"] == "motif3" and isinstance(x["count"], list) and not a.scale:
                # gm_motif (k = 3) takes the reference's formula solver (motif_omp_formula / motif_gpu_formula, src/motif/omp_formula.cc:39-46:
                # the triangles of the oriented graph, wedges derived). The ENUMERATION form (automine_3motif: one bounded intersection of
                # the symmetric lists per edge, tune[6] & 0x10000000) is measured beside it: same counts
                try:
                    f = r.run("motif3e", graphs["motif3"], max(2, min(a.steps, 5)), 1, solo=True)  # (rank 0 alone: no collective here)
                    own_e = own_bytes_device("motif3e", graphs["motif3"])
                    sub["per_edge_variant"] = {"kernel_ms_avg": round(f["kernel_ms_avg"], 4), "n_gpus": 1, "counts_equal": bool(f["count"] == x["count"]),
                                               "own_bytes_per_launch": own_e["bytes"] if own_e else None,
                                               "frac": round(own_e["bytes"] / (f["kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if own_e else None,
                                               "value": round(x["ne_sym"] / (f["kernel_ms_avg"] * 1e-3) / 1e6, 3), "unit": "Medges/s (ne_sym directed edges, the reference's nnz for motif)",
                                               "cpu_baseline": cpu.get("motif3e"),
                                               "note": "gm_motif with tune[6] & 0x10000000: hub rows as hashed sets in LDS, partner lists streamed (gm_hrow.hip)"}
                except Exception as e:  # a report, never a reason to lose the line
                    sub["per_edge_variant"] = {"error": str(e)}
            if x["workload"] in ("diamond", "motif3"):
                # (the default diamond / 3-motif run on structures the FIRST call builds and caches on the caller's symmetric handle --
                # an oriented copy, its topologically renumbered copy where rows are long, task lists with the tasks' own entries, the key
                # stream of the short lists, 4 B of support counter per DAG entry for diamond; none of it is inside kernel_ms)
                sub["first_call_builds"] = ("oriented copy + renumbered copy + task lists / key stream (+ edge supports): setup_ms = orient_ms + relabel_ms + table_ms, "
                                            "end_to_end_ms = graph resident -> first count; about 36 B of HBM per DAG edge beside the symmetric CSR")
            if x["workload"] == "motif3":
                # (`value` divides the reference's nnz for motif -- ne_sym directed edges, src/motif/gpu_base.cu:34,47 -- by the step time; the
                # formula solver itself walks the |E+| = ne_sym / 2 task edges of the oriented graph, like motif_omp_formula's TC pass)
                sub["tasks_of_the_formula_solver"] = x["ne_sym"] // 2
                sub["value_on_formula_tasks"] = round(x["ne_sym"] // 2 / (x["elapsed"] / x["steps"]) / 1e6, 3)
            if x["workload"] == "motif3" and isinstance(x["count"], list):
                # size-independent identity, checked in the run: wedges = sum_v C(d,2) - 3T  (automine_formula.h:2-19)
                bg = graphs["motif3"]
                deg = (bg.rp[1:] - bg.rp[:-1])
                c2 = int((deg * (deg - 1) // 2).sum().item())
                sub["identity_wedges_eq_sumC2_minus_3T"] = bool(x["count"][0] == c2 - 3 * x["count"][1])
            subs.append(sub)
        # the headline workload on a graph whose DAG (1.2 GB) does not fit the 256 MiB Infinity Cache: TC on config 5's R-MAT-24.
        # gm_motif's formula solver launches exactly this kernel on this DAG, so the counter traffic of config 5 is this workload's too.
        tc24 = None
        if not single and "motif3" in graphs and world == 1:
            try:
                rec24 = r.run("tc", graphs["motif3"], max(2, min(a.steps, 10)), 1)
                own24 = own_bytes_device("tc", graphs["motif3"])
                ab24, fl24 = alg_bytes_device("tc", graphs["motif3"], r.lib, rec24["g"])
                rec24.update({"graph": graphs["motif3"].name})
                m3 = next((y for y in subs if y["workload"] == "motif3"), None)
                tc24 = finish_record(rec24, a, world, ab24, fl24, traffic.get("motif3"), traffic_src + " (config 5's pass: the formula 3-motif launches this kernel on this DAG)",
                                     None, None, stream_gbs, own24)
                tc24["count_equals_motif3_triangles"] = bool(m3 is not None and isinstance(m3["count"], list) and m3["count"][1] == rec24["count"])
                tc24["note"] = "TC on R-MAT-24 ef 16: DAG 1.2 GB > 256 MiB Infinity Cache -- an HBM-resident figure for the headline workload"
            except Exception as e:  # a report, never a reason to lose the line
                tc24 = {"error": str(e)}
        head = subs[0]
        out = {
            "metric": "million edges processed/sec + total match count",
            "value": head["value"], "unit": "Medges/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 (vertex ids, offsets; uint64 counts)",
            "data": "file" if head["graph"].startswith("file:") else ("synthetic uniform random graph (torch RNG)" if a.uniform else
                                                                     "synthetic Chung-Lu power-law graph (torch RNG)" if a.powerlaw else
                                                                     "synthetic planted communities + power-law background (torch RNG)" if a.community else
                                                                     "synthetic R-MAT (0.57,0.19,0.19,0.05), SplitMix64 counter stream"),
            "config": {"workload": f"{head['workload']}: {WORKLOADS[head['workload']][3]}", "graph": head["graph"], "nv": head["nv"],
                       "ne_sym": head["ne_sym"], "tasks": head["tasks"], "max_degree": head["max_degree"],
                       "parallelism": f"task-chunk round-robin x{world}, replicated CSR", "input_build_s": head["input_build_s"]},
            "count": head["count"], "matches_per_sec": head["matches_per_sec"], "kernel_ms_avg": head["kernel_ms_avg"],
            "per_gpu_kernel_ms": head["per_gpu_kernel_ms"], "setup_ms": head["setup_ms"], "first_call_ms": head["first_call_ms"], "first_call_runs_ms": head.get("first_call_runs_ms"),
            "roofline": head["roofline"],
        }
        if not single and world == 1 and not a.scale and not a.data_dir and not a.no_standins:
            # Config 2 / 3 name LiveJournal, which is not in the image. Beside the R-MAT stand-in (hub-dominated: max degree
            # 115 k) two more graphs with LiveJournal's |V| and |E| bracket its shape: a flat-degree one (mean oriented list 9,
            # the short-list regime) and a Chung-Lu power-law one with LiveJournal's published maximum degree (20 k).
            extra = []
            # ... and a third with LiveJournal's TRIANGLE DENSITY (planted communities: 288 M triangles; the other two have 0 / 4 M)
            for kind, spec in (("uniform", "4847571,43000000"), ("powerlaw", "4847571,43000000,20000"), ("community", "4847571,43000000")):
                a2 = argparse.Namespace(**vars(a))
                a2.uniform, a2.powerlaw, a2.community = (spec if kind == "uniform" else ""), (spec if kind == "powerlaw" else ""), (spec if kind == "community" else "")
                bg2 = build_graph(a2, r.local_rank, 0, 0)
                tr2 = {}
                if a.traffic == "auto":  # counter traffic of the LiveJournal-sized graphs too (two more passes per graph, the triangle count only)
                    got2, _src2 = measure_traffic(a2, ["tc"], device=r.local_rank, issue=False)
                    tr2 = got2 or {}
                for w in ("tc", "diamond"):
                    rec2 = r.run(w, bg2, a.steps, a.warmup)
                    ab2, fl2 = alg_bytes_device(w, bg2, r.lib, rec2["g"])
                    own2 = own_bytes_device(w, bg2)
                    t2 = rec2["kernel_ms_avg"] * 1e-3
                    extra.append({"workload": w, "graph": bg2.name, "nv": rec2["nv"], "tasks": rec2["tasks"], "max_degree": rec2["max_degree"],
                                  "kernel_ms_avg": round(rec2["kernel_ms_avg"], 4), "value": round(rec2["tasks"] / (rec2["elapsed"] / rec2["steps"]) / 1e6, 3),
                                  "unit": "Medges/s", "count": rec2["count"], "algorithmic_bytes_per_launch": ab2,
                                  "algorithmic_frac": round(ab2 / t2 / 1e9 / HBM_PEAK_GBS, 5) if ab2 and t2 > 0 else None,
                                  "own_bytes_per_launch": own2["bytes"] if own2 else None, "own_streamed_keys": own2["streamed_keys"] if own2 else None,
                                  "frac": round(own2["bytes"] / t2 / 1e9 / HBM_PEAK_GBS, 5) if own2 and t2 > 0 else None,
                                  "compulsory_floor_bytes": fl2, "setup_ms": rec2["setup_ms"],
                                  "traffic": int(tr2[w]["fetch_bytes"] + tr2[w].get("write_bytes", 0.0)) if w in tr2 and "fetch_bytes" in tr2[w] else None,
                                  "traffic_frac": (round((tr2[w]["fetch_bytes"] + tr2[w].get("write_bytes", 0.0)) / t2 / 1e9 / HBM_PEAK_GBS, 5)
                                                   if w in tr2 and "fetch_bytes" in tr2[w] and t2 > 0 else None)})
                bg2.free()
            out["livejournal_standins"] = extra
        if tc24 is not None:
            out["tc_rmat24"] = tc24
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        out["count_matches_cpu"] = head.get("count_matches_cpu")
        if not single:
            out["configs"] = [config1_record(a)] + subs
            out["all_counts_match_cpu"] = all(s.get("count_matches_cpu") is True for s in out["configs"])
        out["rccl_ranks_seen"] = r.ranks_seen
        out["collective_backend"] = (r.backend if r.use_dist else None)
        out["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
        # the full record goes to a file; stdout carries ONE compact line (the driver parses the last line of an 8 KB tail)
        detail = a.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail)), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f)
        except OSError as e:
            print(f"[bench] could not write {detail}: {e}", file=sys.stderr)
            detail = None
        rec = compact_line(out, os.path.relpath(detail, ROOT) if detail and os.path.abspath(detail).startswith(ROOT) else detail)
        for drop in ("", "tc_rmat24", "per_gpu_kernel_ms", "all_counts_match_cpu", "standins"):  # (never reached with the five configs: a guard, not a plan)
            rec.pop(drop, None)
            line = json.dumps(rec, separators=(",", ":"))
            if len(line) < 4000:
                break
    r.close()
    if rank == 0:
        # the LAST thing on stdout, after the communicator is gone and every C buffer of this process is flushed; the other ranks print
        # nothing themselves and get a moment to exit (their RCCL teardown may write through stdio too)
        if r.use_dist and world > 1:
            time.sleep(1.0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
