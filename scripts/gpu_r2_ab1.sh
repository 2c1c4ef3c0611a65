#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
python scripts/ab.py $O/ab_edesc.json e0p0,e1p0,e1p1 \
  'tc_rmat22:--workload;tc;--steps;20;--warmup;3' \
  'tc_uniform:--workload;tc;--uniform;4847571,43000000;--steps;20;--warmup;3' \
  'tc_powerlaw:--workload;tc;--powerlaw;4847571,43000000,20000;--steps;20;--warmup;3' \
  'diamond_rmat22:--workload;diamond;--steps;10;--warmup;2' \
  'diamond_powerlaw:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;10;--warmup;2' \
  'clique4_rmat22:--workload;clique4;--steps;5;--warmup;1' \
  'motif3_rmat24:--workload;motif3;--steps;5;--warmup;1' 2>&1 | tee $O/ab_edesc.log
