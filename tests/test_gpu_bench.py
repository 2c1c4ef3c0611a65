"""bench.py contract (driver-facing): one JSON line with the required fields, roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_small_tc():
    d = run_bench("--scale", "14", "--ef", "8", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Medges/s" and d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c.get("count_matches_gpu", True)
    assert "workload" in d["config"] and "model" not in d["config"]


@pytest.mark.parametrize("workload", ["diamond", "clique4", "motif3", "rectangle", "house"])
def test_bench_other_workloads_run(workload):
    d = run_bench("--workload", workload, "--scale", "12", "--ef", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    assert d["value"] > 0 and d["config"]["workload"].startswith(workload)
