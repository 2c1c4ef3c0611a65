#!/usr/bin/env python3
"""SURVEY 8(f) rows measured: rectangle / house / pentagon (SgL), 5-clique, 4-motif on one R-MAT graph --
kernel ms (HIP events), counter traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE through bench.py's own passes), the slowest kernel
of each workload, and the REFERENCE binary's time on the same graph (oracle/_ref, all host threads, its own Timer; bounded by --ref-timeout).
usage: frows.py [--scale 20] [--ef 16] [--ref-timeout 600] [--out profiles/r05/frows.json] [--workloads rectangle,house,...]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = {  # workload -> (reference binary, extra argv, result pattern, number of result values)
    "rectangle": ("sgl_omp_base", ["rectangle"], r"total_num = (\d+)", 1),
    "house": ("sgl_omp_base", ["house"], r"total_num = (\d+)", 1),
    "pentagon": ("sgl_omp_base", ["pentagon"], r"total_num = (\d+)", 1),
    "clique5": ("clique_omp_base", ["5"], r"num_5-cliques = (\d+)", 1),
    "motif4": ("motif_omp_formula", ["4"], r"pattern \d+: (\d+)", 6),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--ref-timeout", type=int, default=600)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05", "frows.json"))
    ap.add_argument("--workloads", default="rectangle,house,pentagon,clique5,motif4")
    ap.add_argument("--no-ref", action="store_true")
    a = ap.parse_args()
    import oracle as O
    from graphminer_amd.rmat import rmat_csr_device

    threads = O.num_threads()
    tmp = tempfile.mkdtemp(prefix="gm_frows_", dir="/tmp")
    sym, _rp, _ci = rmat_csr_device(a.scale, a.ef, 42, 0)
    host = sym.download()
    prefix = os.path.join(tmp, "graph")
    host.save(prefix)
    sym.free()
    del _rp, _ci
    rows = []
    for w in a.workloads.split(","):
        detail = os.path.join(tmp, f"{w}.json")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--scale", str(a.scale), "--ef", str(a.ef), "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--first-call-repeats", "0", "--detail", detail]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
        row = {"workload": w, "graph": f"rmat_s{a.scale}_ef{a.ef}_seed42"}
        if r.returncode != 0:
            row["error"] = (r.stderr or r.stdout)[-400:]
            rows.append(row)
            continue
        d = json.load(open(detail))
        rf = d["roofline"]
        ks = rf.get("traffic_kernels") or {}
        row.update({"nv": d["config"]["nv"], "ne_sym": d["config"]["ne_sym"], "tasks": d["config"]["tasks"], "count": d["count"],
                    "kernel_ms": d["kernel_ms_avg"], "first_call_ms": d["first_call_ms"], "value_Medges_s": d["value"],
                    "traffic_bytes": rf.get("traffic"), "traffic_GBs": rf.get("traffic_GBs"), "frac_of_8TBs": rf.get("traffic_frac"),
                    "traffic_by_kernel_GB": {k: round((v.get("FETCH_SIZE", 0) * 2048 + v.get("WRITE_SIZE", 0) * 1024) / 1e9, 3) for k, v in ks.items()}})
        if ks:
            row["kernel_with_most_traffic"] = max(row["traffic_by_kernel_GB"], key=row["traffic_by_kernel_GB"].get)
        if not a.no_ref:
            exe, extra, pat, nvals = REF[w]
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread")
            try:
                rr = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe), prefix, *extra], capture_output=True, text=True, env=env, timeout=a.ref_timeout)
                c = [int(x) for x in re.findall(pat, rr.stdout)]
                m = re.search(r"runtime(?: \[[a-z_]+\])? = ([0-9.eE+-]+)", rr.stdout)
                cpu = c[-1] if nvals == 1 else c[-nvals:]
                row["reference"] = {"binary": f"oracle/_ref/{exe} {' '.join(extra)}", "threads": threads, "seconds": float(m.group(1)) if m else None,
                                    "count": cpu, "count_equals_gpu": cpu == d["count"]}
                if m:
                    row["gpu_over_reference"] = round(float(m.group(1)) * 1e3 / d["kernel_ms_avg"], 1)
            except subprocess.TimeoutExpired:
                row["reference"] = {"binary": f"oracle/_ref/{exe} {' '.join(extra)}", "threads": threads, "seconds": None,
                                    "note": f"not finished within {a.ref_timeout} s"}
        rows.append(row)
        print(json.dumps(row), flush=True)
    slowest = max((r for r in rows if "kernel_ms" in r), key=lambda r: r["kernel_ms"], default=None)
    out = {"graph": f"rmat_s{a.scale}_ef{a.ef}_seed42", "rows": rows, "slowest_workload": slowest and slowest["workload"]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
