#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
CASES=('tc_rmat22:--workload;tc;--steps;20;--warmup;3' 'tc_uniform:--workload;tc;--uniform;4847571,43000000;--steps;20;--warmup;3' 'diamond_rmat22:--workload;diamond;--steps;10;--warmup;2' 'clique4_rmat22:--workload;clique4;--steps;5;--warmup;1' 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'tc_rmat24:--workload;tc;--scale;24;--ef;16;--steps;5;--warmup;1')
( echo "== device tables (blocked greedy walk from LDS)"; GM_TABLE_INFO=1 python scripts/ab.py $O/ab_dev.json default "${CASES[@]}" ) 2>&1 | tee $O/ab_setup.log
