#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
CASES=('tc_rmat22:--workload;tc;--steps;20;--warmup;3' 'tc_uniform:--workload;tc;--uniform;4847571,43000000;--steps;20;--warmup;3' 'tc_powerlaw:--workload;tc;--powerlaw;4847571,43000000,20000;--steps;20;--warmup;3' 'diamond_rmat22:--workload;diamond;--steps;10;--warmup;2' 'clique4_rmat22:--workload;clique4;--steps;5;--warmup;1' 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'tc_rmat24:--workload;tc;--scale;24;--ef;16;--steps;5;--warmup;1')
( echo "== device tables (blocked greedy walk)"; python scripts/ab.py $O/ab_dev.json default "${CASES[@]}" ) 2>&1 | tee $O/ab_setup.log
