#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python scripts/ab.py gpurun_out/r2s/a.json default 'tc_rmat22:--workload;tc;--steps;10;--warmup;2' 'tc_uniform:--workload;tc;--uniform;4847571,43000000;--steps;20;--warmup;3' 'tc_powerlaw:--workload;tc;--powerlaw;4847571,43000000,20000;--steps;20;--warmup;3' 'clique4:--workload;clique4;--steps;5;--warmup;1' 'diamond_rmat22:--workload;diamond;--steps;5;--warmup;1' 'motif3f:--workload;motif3f;--steps;3;--warmup;1' 2>&1 | cut -c1-100
