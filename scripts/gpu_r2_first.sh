#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, the default bench line (five configs), the short-list TC stand-in
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 1500 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r2a/bench_default.json") if l.startswith("{")][0])
    print("wall", d["bench_wall_s"], "all_match", d.get("all_counts_match_cpu"))
    for c in d["configs"][1:]:
        r = c["roofline"]
        print(c["workload"], c["graph"], "ms", c["kernel_ms_avg"], "Medges/s", c["value"], "count_ok", c["count_matches_cpu"], "alg_frac", r.get("algorithmic_frac"),
              "traffic_GBs", r.get("traffic_GBs"), "frac", r.get("frac"), "setup", c["setup_ms"], "first", c["first_call_ms"], "cpu", c.get("cpu_baseline", {}).get("value"), c.get("cpu_baseline", {}).get("kind"))
    print("stream", d["roofline"]["stream_ceiling_GBs"], d["roofline"]["traffic_source"])
except Exception as e:
    print("parse failed", e)
PY
timeout 600 python bench.py --workload tc --uniform 4847571,43000000 --steps 20 --warmup 3 --no-cpu-baseline --traffic off > $O/bench_tc_uniform.json 2> $O/bench_tc_uniform.err
python -c "
import json; d=json.loads(open('$O/bench_tc_uniform.json').read()); print('uniform tc ms', d['kernel_ms_avg'], 'frac', d['roofline']['frac'], d['setup_ms'])"
