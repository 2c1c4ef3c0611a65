#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --configs diamond,motif3 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
for c in d['configs'][1:]: print('default-mode', c['workload'], c['kernel_ms_avg'], c['setup_ms'], c['count_matches_cpu'])
for s_ in d.get('livejournal_standins',[]): print(s_['workload'], s_['graph'][:20], s_['kernel_ms_avg'])"
