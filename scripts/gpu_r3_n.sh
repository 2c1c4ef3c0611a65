#!/bin/bash
# round 3, GPU call N: setup times with / without the temp pool
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
for w in tc diamond motif3 clique4; do
  for pool in 1 0; do
    if [ $pool = 0 ]; then export GM_NO_TEMP_POOL=1; else unset GM_NO_TEMP_POOL; fi
    python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --traffic off > $O/${w}_pool$pool.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/${w}_pool$pool.json')); print('$w pool=$pool', {k:round(v,1) for k,v in d['setup_ms'].items()}, 'first_call', round(d['first_call_ms'],1), 'kernel', d['kernel_ms_avg'])"
  done
done
