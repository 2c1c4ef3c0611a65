#!/usr/bin/env python3
"""A/B runner for variant builds of the library (graphminer_amd/variants/lib_<name>.so, selected through GM_LIB_PATH):
runs bench.py in one-workload mode for every (variant, case) pair and prints kernel ms / count per pair.
usage: ab.py <out.json> <variant,variant,...> <case> [<case> ...]     case = label:bench-args (separated by ;)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path, variants, cases = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
res = {}
for case in cases:
    label, args = case.split(":", 1)
    for v in variants:
        env = dict(os.environ)
        if v != "default":
            env["GM_LIB_PATH"] = os.path.join(ROOT, "graphminer_amd", "variants", f"lib_{v}.so")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--traffic", "off", *args.split(";")]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode or not line:
                res[f"{label}/{v}"] = {"error": (r.stderr or r.stdout)[-400:]}
                print(label, v, "ERROR", (r.stderr or r.stdout)[-300:], flush=True)
                continue
            for l in r.stderr.splitlines():
                if l.startswith("["):
                    print("    ", l[:200], flush=True)
            d = json.loads(line[0])
            corner = (d.get("roofline") or {}).get("corner_kernel") or {}
            res[f"{label}/{v}"] = {"kernel_ms": d["kernel_ms_avg"], "ms_per_step": d["ms_per_step"], "count": d["count"],
                                   "corner_ms": corner.get("ms"), "first_call_ms": d.get("first_call_ms")}
            print(f"{label:28s} {v:12s} kernel {d['kernel_ms_avg']:10.4f} ms  corner {corner.get('ms')}  count {d['count']}  first call {d.get('first_call_ms')}", flush=True)
        except Exception as e:
            res[f"{label}/{v}"] = {"error": str(e)}
            print(label, v, "EXC", e, flush=True)
    json.dump(res, open(out_path, "w"), indent=1)
