"""bench.py contract (driver-facing): one JSON line with the required fields, roofline and cpu_baseline objects, the five
BASELINE configs as sub-records, and the torch.distributed (RCCL) path exercised with one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_roofline(r):
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] > 0
    if r.get("frac") is not None:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] <= 1.0
    assert r["compulsory_floor_bytes"] > 0 and r["stream_ceiling_GBs"] > 1000


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
    assert cfg[1]["count"] == cfg[4]["count"][1]  # triangles: TC kernel on the DAG == 3-motif kernel on the symmetric graph


def test_bench_distributed_path_with_one_rank():
    """GM_BENCH_FORCE_DIST=1: torch.distributed 'nccl' (= RCCL) init, the all-reduce of the counts, barriers, max-over-ranks
    timing and teardown all run -- with world size 1, so that the driver's first 8-GPU run is not this code's first execution"""
    d = run_bench("--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off",
                  env={"GM_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert d["n_gpus"] == 1 and len(d["configs"]) == 5
    ref = run_bench("--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off")
    assert [c["count"] for c in d["configs"][1:]] == [c["count"] for c in ref["configs"][1:]]
    assert d["per_gpu_kernel_ms"]["max"] >= d["per_gpu_kernel_ms"]["mean"] > 0


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


@pytest.mark.parametrize("workload", ["diamond", "clique4", "motif3", "rectangle", "house"])
def test_bench_other_workloads_run(workload):
    d = run_bench("--workload", workload, "--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--traffic", "off")
    assert d["value"] > 0 and d["config"]["workload"].startswith(workload)
