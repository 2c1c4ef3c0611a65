#!/bin/bash
# round 3, GPU call D: whole GPU suite + the default bench line (own-algorithm roofline, PMC traffic, CPU baselines)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
(time timeout 1800 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3d/bench_default.json'))
print(d['value'], d['ms_per_step'], d['bench_wall_s'], d['all_counts_match_cpu'])
for c in d['configs'][1:]:
    r=c['roofline']; print(c['workload'], c['kernel_ms_avg'], 'frac', r.get('frac'), 'alg', r.get('algorithmic_frac'), 'traffic_frac', r.get('traffic_frac'), 'setup', sum(c['setup_ms'].values()), c.get('count_matches_cpu'))
PY
