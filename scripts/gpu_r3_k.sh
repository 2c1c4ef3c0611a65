#!/bin/bash
# round 3, GPU call K: share scaling after the fixes (4 chunks / workgroup for shares, giant chunk target, small build batches), big graph test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
S="python scripts/sim_scale.py --reps 3"
( $S --workload tc --scale 22 --ef 10
  $S --workload diamond --scale 22 --ef 10
  $S --workload clique4 --scale 22 --ef 28
  $S --workload motif3 --scale 24 --ef 16
  $S --workload diamond --scale 24 --ef 16 --worlds 1,8 ) 2>&1 | grep -v amdgpu.ids > $O/sim_scale_one_gpu.txt
cat $O/sim_scale_one_gpu.txt
(time timeout 1700 python -m pytest tests/test_gpu_fullsize.py -q -x -k "2e31" -rs -s) 2>&1 | grep -v amdgpu.ids | tail -8 > $O/pytest_big.log; cat $O/pytest_big.log
