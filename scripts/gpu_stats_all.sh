#!/bin/bash
# rocprofv3 --kernel-trace --stats for every bench workload (summaries only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$PWD; export TMPDIR=/tmp; KEEP=$REPO/gpurun_out/stats_all; rm -rf $KEEP; mkdir -p $KEEP
cd /tmp
for w in tc diamond clique4 motif3 motif3f rectangle house pentagon clique5; do
  rm -rf /tmp/st_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$w -o t -- python $REPO/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $KEEP/bench_$w.log 2>&1
  head -4 /tmp/st_$w/t_kernel_stats.csv | cut -c1-220 > $KEEP/kernel_stats_$w.csv
  grep -h "^{" $KEEP/bench_$w.log | tail -1 > $KEEP/bench_line_$w.json
  cat $KEEP/bench_line_$w.json | python $REPO/scripts/short.py
done
