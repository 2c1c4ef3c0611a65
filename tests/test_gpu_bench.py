"""bench.py contract (driver-facing): one JSON line with the required fields, roofline and cpu_baseline objects, the five
BASELINE configs as sub-records, and the torch.distributed (RCCL) path exercised with one rank."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


LINE_KEYS = REQUIRED + ["count", "rccl_ranks_seen", "detail"]
ROOF_KEYS = ["bound", "achieved", "peak", "unit", "frac", "traffic", "frac_basis", "kernel", "stream_ceiling_GBs"]


def check_line(line, raw):
    """the stdout line the driver parses: short, self-contained, the contract's keys"""
    assert len(raw) < 4096, len(raw)
    for k in LINE_KEYS:
        assert k in line, k
    for k in ROOF_KEYS:
        assert k in line["roofline"], k
    if line["cpu_baseline"] is not None:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
    assert "workload" in line["config"] and "model" not in line["config"]


def run_bench(*args, env=None, launcher=None):
    """runs bench.py; returns the FULL record (the --detail file) with the parsed stdout line under "_line" """
    import tempfile

    with tempfile.TemporaryDirectory(prefix="gm_bench_test_") as tmp:
        detail = os.path.join(tmp, "detail.json")
        cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py"), *args, "--detail", detail]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=dict(os.environ, **(env or {})))
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and out.stdout.rstrip().splitlines()[-1] == lines[0], out.stdout[-2000:]  # ONE line, and it is the last
        line = json.loads(lines[0])
        check_line(line, lines[0])
        d = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "count"):  # the line is a projection of the record
        assert line[k] == d[k], k
    assert line["roofline"]["frac"] == d["roofline"]["frac"] and line["roofline"]["traffic"] == d["roofline"]["traffic"]
    d["_line"] = line
    return d


def check_roofline(r):
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] > 0
    if r.get("frac") is not None:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
    if r.get("traffic_frac") is not None:  # frac = counter traffic / kernel time / peak whenever the PMC pass succeeded
        assert r["frac"] == r["traffic_frac"] and abs(r["traffic_GBs"] - r["achieved"]) < 1e-6 and "counter traffic" in r["frac_basis"]
    elif "own_bytes_per_launch" in r:  # no counters in this run: the own-algorithm bytes, and the basis says so
        assert r["frac"] == r["own_frac"] and abs(r["own_GBs"] - r["achieved"]) < 1e-6 and r["frac_basis"].startswith("NO counter pass")
    if "own_bytes_per_launch" in r:
        assert r["own_bytes_per_launch"] >= 4 * r["own_streamed_keys_per_launch"] >= 0 and 0 < r["own_frac"] <= 1.0
    if "algorithmic_frac" in r:
        assert r["frac_8d_valid"] == (r["algorithmic_frac"] <= 1.0)
    assert r["compulsory_floor_bytes"] > 0 and r["stream_ceiling_GBs"] > 1000
    if r.get("traffic_frac") is not None:  # counter traffic of THIS workload's launches only (two workloads may share a kernel name:
        assert 0 < r["traffic_frac"] <= 1.0  # the traffic pass separates them by marker dispatches) -- it cannot exceed the peak


def test_bench_line_small_tc():
    d = run_bench("--workload", "tc", "--scale", "14", "--ef", "8", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--traffic", "off")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Medges/s" and d["value"] > 0 and d["ms_per_step"] > 0
    check_roofline(d["roofline"])
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c.get("count_matches_gpu", True)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert set(d["setup_ms"]) == {"orient_ms", "table_ms", "bitmap_ms", "relabel_ms", "other_ms"} and d["setup_ms"]["table_ms"] > 0


def test_bench_last_line_is_short():
    """the default mode's stdout: exactly one JSON line < 4 KB with roofline + cpu_baseline and a summary per config (r04's 24 KB line
    did not parse at the driver)"""
    d = run_bench("--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--traffic", "off")
    line = d["_line"]
    assert [c["id"] for c in line["configs"]] == [1, 2, 3, 4, 5]
    for c, full in zip(line["configs"][1:], d["configs"][1:]):
        assert c["workload"] == full["workload"] and c["count"] == full["count"] and c["kernel_ms"] == full["kernel_ms_avg"]
        assert c["count_ok"] is True
    assert line["all_counts_match_cpu"] is True and line["rccl_ranks_seen"] == 1
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["count_matches_gpu"] is True


def test_bench_default_mode_carries_the_five_configs():
    """the driver's command line, shrunk to R-MAT-14: five sub-records, every count checked against a CPU count of the run
    (the oracle samples have stride 1 at this size), counter traffic measured by the rocprofv3 passes of the same run"""
    d = run_bench("--scale", "14", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "20")
    for k in REQUIRED:
        assert k in d, k
    cfg = d["configs"]
    assert [c["id"] for c in cfg] == [1, 2, 3, 4, 5]
    assert cfg[0]["count_matches_cpu"] is True and cfg[0]["gpu_count"] == 1166
    assert [c["workload"] for c in cfg[1:]] == ["tc", "diamond", "clique4", "motif3"]
    assert d["value"] == cfg[1]["value"] and d["roofline"] == cfg[1]["roofline"]
    for c in cfg[1:]:
        assert c["value"] > 0 and c["kernel_ms_avg"] > 0 and c["tasks"] > 0
        check_roofline(c["roofline"])
        assert c["roofline"]["traffic"] and c["roofline"]["traffic"] > 0, c["roofline"]["traffic_source"]
        b = c["cpu_baseline"]
        assert b["value"] > 0 and b["cores"] >= 1
        if b["kind"] == "port":  # whole graph at this size: the oracle's count must equal the GPU's
            assert b["stride"] == 1
            assert b["sample_count"] == c["count"], c["workload"]
        else:
            assert b["count_matches_gpu"] is True
    assert cfg[4]["identity_wedges_eq_sumC2_minus_3T"] is True
    assert cfg[4]["tasks_of_the_formula_solver"] * 2 == cfg[4]["tasks"] and all("end_to_end_ms" in c for c in cfg[1:])
    t24 = d["tc_rmat24"]  # (the headline workload on config 5's graph; at this size it is config 2's graph again)
    assert t24["workload"] == "tc" and t24["count"] == cfg[4]["count"][1] and t24["count_equals_motif3_triangles"] is True
    check_roofline(t24["roofline"])
    assert cfg[1]["count"] == cfg[4]["count"][1]  # triangles: TC kernel on the DAG == 3-motif kernel on the symmetric graph


def test_bench_distributed_path_with_one_rank():
    """GM_BENCH_FORCE_DIST=1: torch.distributed 'nccl' (= RCCL) init, the all-reduce of the counts, barriers, max-over-ranks
    timing and teardown all run -- with world size 1, so that the driver's first 8-GPU run is not this code's first execution"""
    d = run_bench("--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "5",
                  env={"GM_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert d["n_gpus"] == 1 and len(d["configs"]) == 5
    ref = run_bench("--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "5")
    assert [c["count"] for c in d["configs"][1:]] == [c["count"] for c in ref["configs"][1:]]
    assert d["per_gpu_kernel_ms"]["max"] >= d["per_gpu_kernel_ms"]["mean"] > 0

    # the distributed line is as complete as the single-process one: same keys at every level that matters
    def keys(x):
        return {k for k in x if k not in ("carried_from",)}

    assert keys(ref) <= keys(d), keys(ref) - keys(d)
    assert keys(ref["roofline"]) <= keys(d["roofline"]), keys(ref["roofline"]) - keys(d["roofline"])
    assert "cpu_baseline" in d and d["cpu_baseline"]["value"] > 0
    for cr, cd in zip(ref["configs"][1:], d["configs"][1:]):
        assert keys(cr) <= keys(cd), (cr["workload"], keys(cr) - keys(cd))
        assert keys(cr["roofline"]) <= keys(cd["roofline"]), (cr["workload"], keys(cr["roofline"]) - keys(cd["roofline"]))
        assert cd["roofline"]["frac"] is not None and 0 < cd["roofline"]["frac"] <= 1
        assert cd["roofline"]["traffic"] and cd["roofline"]["traffic"] > 0, cd["roofline"]["traffic_source"]
        assert cd["cpu_baseline"]["value"] > 0
        assert cd["per_gpu_kernel_ms"]["skew_max_over_mean"] >= 1.0


def test_bench_traffic_of_a_rank_share_on_one_gpu():
    """what rank 0 does at N > 1: counter traffic of ITS share, measured by a child process on one GPU (rank / world are launch arguments)"""
    sys.path.insert(0, ROOT)
    import argparse

    import bench

    a = argparse.Namespace(seed=42, scale=14, ef=8, graph="", data_dir="", uniform="", powerlaw="", tune="", policy=0, share_rank=0, share_world=1)
    whole, src = bench.measure_traffic(a, ["tc", "motif3", "diamond"])
    assert whole, src
    share, src = bench.measure_traffic(a, ["tc", "motif3", "diamond"], share=(0, 4))
    assert share and "share of rank 0 of 4" in src, src
    for w in ("tc", "motif3"):
        assert 0 < share[w]["fetch_bytes"] < whole[w]["fetch_bytes"], (w, share[w], whole[w])
    assert share["diamond"]["fetch_bytes"] > 0 and whole["diamond"]["fetch_bytes"] > 0
    # (round 4: a rank's share of the diamond is a share of the SAME triangle pass -- gm_diamond_support_partial -- not another algorithm)
    assert any("gm::sup_kernel" in k for k in whole["diamond"]["kernels"]) and any("gm::sup_kernel" in k for k in share["diamond"]["kernels"])
    assert share["diamond"]["fetch_bytes"] < whole["diamond"]["fetch_bytes"]


def test_algorithmic_bytes_on_the_device_equal_the_oracle():
    """bench.py computes SURVEY 8(d)'s algorithmic bytes with torch on the GPU; the oracle's exact one-pass figures agree"""
    sys.path.insert(0, ROOT)
    import torch

    import bench
    import oracle as O
    from graphminer_amd import _lib
    from graphminer_amd.rmat import rmat_csr_device

    sym, rp, ci = rmat_csr_device(12, 8, 7, 0)
    bg = bench.BenchGraph(sym, rp, ci, "t", 0.0)
    h = sym.download()
    osym = O.OGraph(h.row_ptr, h.col_idx)
    odag = O.orient(osym)
    lib = _lib.load()
    assert bench.alg_bytes_device("tc", bg, lib, bg.dag())[0] == O.alg_bytes("tc", odag)
    assert bench.alg_bytes_device("diamond", bg, lib, bg.sym)[0] == O.alg_bytes("diamond", osym)
    assert bench.alg_bytes_device("motif3", bg, lib, bg.sym)[0] == O.alg_bytes("motif3", osym)
    assert bench.alg_bytes_device("clique4", bg, lib, bg.dag())[0] == O.alg_bytes("clique4", odag)
    assert bench.alg_bytes_device("tc", bg, lib, bg.dag())[1] == 8 * (odag.nv + 1) + 4 * odag.ne
    torch.cuda.synchronize()


def test_own_algorithm_bytes_equal_a_brute_force_count():
    """roofline.frac's numerator (bench.own_bytes_device) against plain numpy loops over the edges of a small graph"""
    sys.path.insert(0, ROOT)
    import numpy as np

    import bench
    from graphminer_amd.rmat import rmat_csr_device

    sym, rp, ci = rmat_csr_device(11, 500, 3, 0)  # dense enough for long DAG rows (the library renumbers it), wide vertices (d+ > 256: gathered rows) and symmetric rows beyond 128
    bg = bench.BenchGraph(sym, rp, ci, "t", 0.0)
    h = sym.download()
    hrp, hci = h.row_ptr, h.col_idx
    nv, deg = hrp.size - 1, np.diff(hrp)
    # symmetric-graph patterns
    kd = km = 0
    trim_min = bench.kernel_constants()["motif_trim_min_list"]
    rk = np.empty(nv, dtype=np.int64)  # the enumeration runs on the copy numbered by DESCENDING degree: new id = nv - 1 - rank in (degree, id)
    rk[np.lexsort((np.arange(nv), deg))] = np.arange(nv)
    new = nv - 1 - rk
    for u in range(nv):
        for v in hci[hrp[u]:hrp[u + 1]]:
            if v >= u:
                continue
            a, b = deg[u], deg[v]
            u_longer = a > b or (a == b and u > v)
            s, n = (v, b) if u_longer else (u, a)
            kd += n
            # (same edge under the new numbering: the longer row hosts -- ties: the larger NEW id -- and the streamed list keeps its keys below max)
            nu, nw = new[u], new[v]
            m_longer = a > b or (a == b and nu > nw)
            ms, mn = (v, b) if m_longer else (u, a)
            km += int((new[hci[hrp[ms]:hrp[ms + 1]]] < max(nu, nw)).sum()) if mn >= trim_min else mn
    ne = hci.size
    assert bench.own_bytes_device("diamond", bg, 2)["bytes"] == 4 * kd + 12 * ne + 8 * (nv + 1)  # (several ranks: one intersection per edge)
    assert bench.own_bytes_device("motif3e", bg)["bytes"] == 4 * km + 12 * ne + 8 * (nv + 1) and km < kd  # (the enumeration kernels)
    # DAG patterns
    d = bg.dag().download()
    drp, dci = d.row_ptr, d.col_idx
    dp = np.diff(drp)
    kt = kc = tasks = 0
    rank = np.empty(nv, dtype=np.int64)  # the library's topological numbering of the DAG: ids ascending in (symmetric degree, id)
    rank[np.lexsort((np.arange(nv), deg))] = np.arange(nv)
    trim = float((dp.astype(np.float64) ** 2).sum()) / dci.size >= bench.kernel_constants()["topo_min_mean_row"]  # (topo_view: renumbered and trimmed only where rows are long)
    assert trim, "the graph must be one the library renumbers"
    K = bench.kernel_constants()
    core_base = nv - min(nv, K["core_h_default"])  # (new ids; the whole of this small graph lies in the hub core)
    gathered = core_rows = 0
    for u in range(nv):
        row = dci[drp[u]:drp[u + 1]]
        srow = np.sort(rank[row])  # N+(u) under the library's numbering
        wide = dp[u] * ((dp[u] + 31) // 32) > K["wide_min_words"] and dp[u] <= K["cb_max_deg"]
        for i, v in enumerate(row[np.argsort(rank[row])]):
            tail = dp[u] - i - 1
            kt += dp[v] if tail >= dp[v] else tail  # the shorter stream (no row beyond 2048 entries here)
            if K["cb_min_deg"] <= dp[u] <= K["cb_max_deg"]:
                if wide and rank[v] >= core_base:  # gathered from the core bitmap: the distinct words of the columns beyond i
                    core_rows += 1
                    gathered += len(set(((srow[i + 1:] - core_base) >> 5).tolist()))
                else:
                    tasks += 1
                    kc += tail if dp[v] > tail and dp[v] <= K["cb_max_deg"] else dp[v]
    assert core_rows > 0, "the graph must have wide vertices for this check to cover the gathered rows"
    assert dp.max() > 64 and tasks > 0 and kt < int(np.minimum(dp[np.repeat(np.arange(nv), dp)], dp[dci]).sum()), "the graph must have long DAG rows for this check to mean something"
    nd = dci.size
    assert bench.own_bytes_device("tc", bg)["bytes"] == 4 * int(kt) + 12 * nd + 8 * (nv + 1)
    assert bench.own_bytes_device("motif3", bg) == bench.own_bytes_device("tc", bg)  # gm_motif, k = 3: the triangles of the DAG + a closed form
    # diamond on one GPU: edge supports from the triangles of the DAG (gm_sup.hip).  Round 5: an IN-EDGE task (the target hosts, the tail of
    # N+(u) is streamed) with a tail of >= `lmin` keys reports its streamed edges as a bit mask of the tail -- ceil(tail / 64) 64-bit words,
    # three spare ones behind a list of >= long_list keys -- every other task by one atomic per match
    from graphminer_amd import _lib as L

    lmin_info = (C.c_int64 * 4)()
    own_d = bench.own_bytes_device("diamond", bg)
    L.check(L.load().gm_diamond_support_info(bg.sym.handle, lmin_info), "gm_diamond_support_info")
    lmin, long_list = int(lmin_info[3]), K["long_list"]
    tri = atomics = words = 0
    nbr = [set(rank[dci[drp[u]:drp[u + 1]]].tolist()) for u in range(nv)]
    for u in range(nv):
        row = dci[drp[u]:drp[u + 1]]
        for i, v in enumerate(row[np.argsort(rank[row])]):
            t = len(nbr[u] & nbr[v])
            tri += t
            tail = dp[u] - i - 1
            masked = tail < dp[v] and tail >= lmin  # (v hosts: the shorter stream; no row beyond the stage here)
            if masked:
                words += (tail + 63) // 64 + (3 if tail >= long_list else 0)
            else:
                atomics += t
    assert words > 0 and 0 < atomics < tri, "the graph must exercise masked and unmasked tasks"
    assert (int(lmin_info[0]), int(lmin_info[1])) == (words, atomics)
    assert own_d["bytes"] == 4 * int(kt) + 12 * nd + 8 * (nv + 1) + 20 * nd + 4 * atomics + 16 * words + 8 * nd
    assert own_d["parts"]["triangles"] == tri
    own = dp[(dp >= bench.kernel_constants()["cb_min_deg"]) & (dp <= bench.kernel_constants()["cb_max_deg"])].astype(np.int64)
    arena = int((own * ((own + 31) // 32)).sum())
    # round 6: the rows gathered from the hub core go through the BLOCKED gather (csrc/gm_cgather.hip): (vertex, block of core rows) units in
    # block order, each reading a 16-byte record, 4 B per row and its 16-bit column table from its first row's tile pair on; work items of
    # ~2^20 probes, one 64 KB block image each -- restated here from the block geometry of tests/test_corner_blocks.py
    from test_corner_blocks import geometry

    core_h = nv - core_base
    _words, bid, _rowbase, _blk, _total = geometry(core_h, core_base & 31)
    wide_v = [u for u in range(nv) if dp[u] * ((dp[u] + 31) // 32) > K["wide_min_words"] and dp[u] <= K["cb_max_deg"]]
    wide_v.sort(key=lambda u: (-int(dp[u]), u))  # the plan's slot order: longest rows first, stable
    units = []  # (block, slot, first row index, rows, d)
    for slot, u in enumerate(wide_v):
        srow = np.sort(rank[dci[drp[u]:drp[u + 1]]])
        dd = len(srow)
        i = int(np.searchsorted(srow, core_base))
        while i < dd:
            b0 = bid[srow[i] - core_base]
            j = i
            while j < dd and bid[srow[j] - core_base] == b0:
                j += 1
            units.append((int(b0), slot, i, j - i, dd))
            i = j
    units.sort(key=lambda t: (t[0], t[1], t[2]))
    unit_bytes = sum(16 + 4 * r + 256 * max(((dd + 63) // 64 + 1) // 2 - ((i0 + 1) >> 7), 0) for _b, _s, i0, r, dd in units)
    items, cum, prev = 0, 0, None
    for b0, _s, i0, r, dd in units:
        cost = r * (dd - 1 - i0) - r * (r - 1) // 2 + 64 * r + 256
        key = (b0, cum >> 20)
        items += key != prev
        prev = key
        cum += cost
    own_c = bench.own_bytes_device("clique4", bg)
    assert own_c["parts"]["blocked_gather"] == {"units": len(units), "unit_bytes": unit_bytes, "items": items, "blocks": len(_blk)}
    assert own_c["parts"]["core_words_gathered_x4_row_major"] == 4 * gathered  # (the row-major gather of rounds 4 - 5, kept for comparison)
    assert own_c["bytes"] == 4 * int(kc) + 16 * tasks + 4 * nd + 16 * (nv + 1) + 8 * arena + unit_bytes + 65536 * items
    bg.free()


@pytest.mark.parametrize("workload", ["diamond", "clique4", "motif3", "motif3e", "rectangle", "house"])
def test_bench_other_workloads_run(workload):
    d = run_bench("--workload", workload, "--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off")
    assert d["value"] > 0 and d["config"]["workload"].startswith(workload)


def test_scale_dryrun_two_ranks_on_one_gpu():
    """scripts/scale_dryrun.sh: the driver's N > 1 launch line with two ranks on GPU 0 over gloo -- rank shares, the summed counts, the
    per-rank kernel times, the CPU-baseline record and the rank-0-only counter passes (the other rank waits at the barrier) all run, so the
    first 8-GPU job is not the first execution of any of it"""
    import tempfile

    with tempfile.TemporaryDirectory(prefix="gm_bench_test_") as tmp:
        out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "scale_dryrun.sh"), "2", "--scale", "13", "--ef", "8", "--steps", "2", "--warmup", "1",
                              "--cpu-seconds", "3", "--detail", os.path.join(tmp, "d.json")], capture_output=True, text=True, timeout=1500, cwd=ROOT,
                             env=dict(os.environ, MASTER_PORT="29547"))
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]  # rank 0 alone prints
        check_line(json.loads(lines[0]), lines[0])
        assert json.loads(lines[0])["rccl_ranks_seen"] == 2
        d = json.load(open(os.path.join(tmp, "d.json")))
    ref = run_bench("--scale", "13", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "3", "--traffic", "off")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert [c["count"] for c in d["configs"][1:]] == [c["count"] for c in ref["configs"][1:]]  # the two shares add up to the whole
    assert d["all_counts_match_cpu"] is True
    for c in d["configs"][1:]:
        assert len(c["per_gpu_kernel_ms"]["all"]) == 2 and min(c["per_gpu_kernel_ms"]["all"]) > 0
        assert c["cpu_baseline"]["value"] > 0
        assert c["roofline"]["traffic"] and c["roofline"]["traffic"] > 0, c["roofline"]["traffic_source"]
        assert "share of rank 0 of 2" in c["roofline"]["traffic_source"]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it: bench.py re-execs under torch.distributed.run. On this
    one-GPU box through GM_BENCH_ONE_GPU=1 (both ranks on GPU 0, gloo); without that switch the same command must refuse clearly."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    d = run_bench("--gpus", "2", "--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2", "--traffic", "off",
                  env=dict(env, GM_BENCH_ONE_GPU="1"))
    assert d["n_gpus"] == 2 and d["_line"]["rccl_ranks_seen"] == 2 and d["collective_backend"] == "gloo"
    assert d["all_counts_match_cpu"] is True
    for c in d["configs"][1:]:
        assert len(c["per_gpu_kernel_ms"]["all"]) == 2
    import torch

    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "12"], capture_output=True, text=True, timeout=300,
                             cwd=ROOT, env=env)
        assert out.returncode != 0 and "only 1 HIP device" in out.stderr and not out.stdout.strip()
