"""bench.py's stdout line (CPU-only checks): the compact projection of a full record stays under 4 KB and keeps the contract's keys --
fed with round 4's own 24 KB record (profiles/r04/bench_default_line.json), the line the driver could not parse."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_compact_line_of_the_round_4_record():
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r04", "bench_default_line.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    raw = json.dumps(line, separators=(",", ":"))
    assert len(raw) < 4096, len(raw)
    for k in REQUIRED:
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_basis", "kernel"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["roofline"]["traffic"] == full["roofline"]["traffic"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert [c["id"] for c in line["configs"]] == [1, 2, 3, 4, 5]
    assert [c["workload"] for c in line["configs"][1:]] == ["tc", "diamond", "clique4", "motif3"]
    assert line["value"] == full["value"] and line["count"] == full["count"]
    assert "model" not in line["config"] and line["config"]["workload"].startswith("tc")


def test_compact_line_of_the_round_5_record_carries_the_corner_kernel():
    """round 5: the headline's launch has a hub-corner kernel on the matrix cores (csrc/gm_ctc.hip) -- the line prices the streamed kernels
    against the HBM roofline and carries the MFMA part beside it, and still fits 4 KB"""
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_default_detail.json")))
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    raw = json.dumps(line, separators=(",", ":"))
    assert len(raw) < 4096, len(raw)
    for k in REQUIRED:
        assert k in line, k
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] <= 1 and roof["streamed_kernels_ms"] < line["kernel_ms_avg"]
    ck = roof["corner_kernel"]
    assert ck["bound"] == "mfma" and 0 < ck["frac"] <= 1 and abs(ck["ms"] + roof["streamed_kernels_ms"] - line["kernel_ms_avg"]) < 0.02
    assert all(c["count_ok"] for c in line["configs"])


def test_full_size_cpu_records_are_on_the_bench_graphs():
    import bench

    c4 = bench.full_size_cpu("clique4", "rmat_s22_ef28_seed42", 110_000_000)
    m3 = bench.full_size_cpu("motif3", "rmat_s24_ef16_seed42", 520_000_000)
    assert c4 and c4["seconds"] > 600 and c4["threads"] == 128 and c4["value"] > 0
    assert m3 and m3["seconds"] > 30 and "motif_omp_formula" in m3["binary"]  # the formula solver's own CPU counterpart, on R-MAT-24
    assert bench.full_size_cpu("tc", "rmat_s22_ef10_seed42", 1) is None  # (timed in every run instead)


def test_self_launch_is_a_no_op_under_a_launcher(monkeypatch):
    import argparse

    import bench

    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.self_launch(argparse.Namespace(gpus=8, traffic_worker=False)) is None  # torch.distributed.run already started the ranks
    monkeypatch.delenv("WORLD_SIZE")
    assert bench.self_launch(argparse.Namespace(gpus=1, traffic_worker=False)) is None
