#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "diamond or motif or hub or random_graphs or large_rmat or rows_longer" 2>&1 | tail -8 | tee $O/pytest_sym.log
OLD='--tune;0,0,0,0,0,0,524288'
python scripts/ab.py $O/ab_sym.json default \
  'diamond_rmat22:--workload;diamond;--steps;10;--warmup;2' "diamond_rmat22_old:--workload;diamond;--steps;10;--warmup;2;$OLD" \
  'diamond_rmat20:--workload;diamond;--scale;20;--ef;16;--steps;10;--warmup;2' "diamond_rmat20_old:--workload;diamond;--scale;20;--ef;16;--steps;10;--warmup;2;$OLD" \
  'diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1' "diamond_rmat24_old:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1;$OLD" \
  'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' "motif3_rmat24_old:--workload;motif3;--steps;3;--warmup;1;$OLD" \
  'motif3_rmat22:--workload;motif3;--scale;22;--ef;16;--steps;5;--warmup;1' "motif3_rmat22_old:--workload;motif3;--scale;22;--ef;16;--steps;5;--warmup;1;$OLD" \
  'diamond_powerlaw:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;10;--warmup;2' "diamond_powerlaw_old:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;10;--warmup;2;$OLD" 2>&1 | tee $O/ab_sym.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sym -o trace -- python $OLDPWD/bench.py --workload motif3 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; cd $OLDPWD
find /tmp/prof_sym -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_motif3.csv \;
head -6 $O/kernel_stats_motif3.csv | cut -c1-150
