#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "clique or complete or big_rows or random_graphs or large_rmat" 2>&1 | tail -8 | tee $O/pytest_clique.log
C='clique4_rmat22:--workload;clique4;--steps;3;--warmup;1'
( GM_WIDE_PROFILE=1 python scripts/ab.py $O/a1.json default "$C"
  python scripts/ab.py $O/a2.json default "$C" 'clique4_rmat22_old:--workload;clique4;--steps;3;--warmup;1;--tune;0,0,0,0,0,0,262144' 'clique4_rmat20:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1' 'clique4_rmat20_old:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,262144'
) 2>&1 | tee $O/ab_wide3.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wide -o trace -- python $OLDPWD/bench.py --workload clique4 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; cd $OLDPWD
find /tmp/prof_wide -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_clique4.csv \;
head -8 $O/kernel_stats_clique4.csv | cut -c1-150
