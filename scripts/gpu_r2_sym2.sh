#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
for spec in "diamond 22 10" "diamond 24 16" "motif3 24 16"; do
  set -- $spec
  echo "=== $1 rmat$2 ef$3"
  GM_TABLE_INFO=1 python bench.py --workload $1 --scale $2 --ef $3 --steps 2 --warmup 1 --no-cpu-baseline --traffic off 2>&1 >/dev/null | grep "^\[table\]"
  cd /tmp && rm -rf /tmp/prof_s && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o trace -- python $OLDPWD/bench.py --workload $1 --scale $2 --ef $3 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; cd $OLDPWD
  find /tmp/prof_s -name "*kernel_stats.csv" -exec cat {} \; | grep mine_kernel | cut -d, -f1-4
done 2>&1 | tee $O/sym_classes_info.log
