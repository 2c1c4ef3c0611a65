#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clique or complete or big_rows or random_graphs or large_rmat" 2>&1 | tail -8 | tee $O/pytest_clique.log
python scripts/ab.py $O/ab_wide.json default \
  'clique4_rmat22_wide:--workload;clique4;--steps;5;--warmup;1' \
  'clique4_rmat22_old:--workload;clique4;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,262144' \
  'clique4_rmat22_nocount:--workload;clique4;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,1' \
  'clique4_rmat20_wide:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1' \
  'clique4_rmat20_old:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,262144' 2>&1 | tee $O/ab_wide.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wide -o trace -- python $OLDPWD/bench.py --workload clique4 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; cd $OLDPWD
find /tmp/prof_wide -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_clique4_wide.csv \;
head -8 $O/kernel_stats_clique4_wide.csv | cut -c1-160
