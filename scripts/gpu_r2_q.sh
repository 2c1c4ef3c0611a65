#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2r
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clique" 2>&1 | tail -3
python scripts/ab.py gpurun_out/r2r/a.json default 'clique4_rmat22:--workload;clique4;--steps;5;--warmup;1' 'clique4_rmat20:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1' 2>&1 | tee gpurun_out/r2r/clique.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o trace -- python $OLDPWD/bench.py --workload clique4 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; cd $OLDPWD
find /tmp/prof_c -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2r/kernel_stats_clique4.csv \;
grep "clique\|mine_kernel" gpurun_out/r2r/kernel_stats_clique4.csv | cut -d, -f1-4 | cut -c1-120
