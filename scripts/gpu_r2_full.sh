#!/bin/bash
# whole -m gpu suite + default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r2j}; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -16 $O/pytest_gpu.log
( time timeout 1700 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 600 $O/bench_default.err
python - $O <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1] + "/bench_default.json") if l.startswith("{")][0])
    print("wall", d["bench_wall_s"], "all_match", d.get("all_counts_match_cpu"), "stream", d["roofline"]["stream_ceiling_GBs"])
    for c in d["configs"][1:]:
        r = c["roofline"]; b = c.get("cpu_baseline", {})
        print(c["workload"], c["graph"], "ms", c["kernel_ms_avg"], "Medges/s", c["value"], "ok", c["count_matches_cpu"], "alg_frac", r.get("algorithmic_frac"),
              "traffic_GB", round((r.get("traffic") or 0) / 1e9, 1), "tr_GBs", r.get("traffic_GBs"), "frac", r.get("frac"), "setup", c["setup_ms"], "cpu", b.get("value"), b.get("kind"), b.get("seconds"), b.get("count_matches_gpu"))
except Exception as e:
    print("parse failed", e)
PY
