#!/bin/bash
# Round-end evidence on the GPU box: the full GPU suite, the default bench line, and the per-config rocprofv3 / PMC summaries.
# usage: bash scripts/round_end.sh <out dir under gpurun_out/>   (copy what is to be judged into profiles/rNN/)
out=gpurun_out/${1:-round_end}
mkdir -p "$out"
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests -m gpu -q -x ) > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?"; tail -3 "$out/pytest_gpu.log"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$out/bench_default_detail.json" > "$out/bench_default_line.json" 2> "$out/bench_default.err"
echo "bench rc=$?"; python - "$out/bench_default_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("bench_wall_s"))
for c in d.get("configs", []):
    print(" ", c.get("name"), c.get("kernel_ms_avg"), (c.get("roofline") or {}).get("frac"), c.get("setup_ms"), c.get("count_matches_cpu"))
PY
