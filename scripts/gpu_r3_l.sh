#!/bin/bash
# round 3, GPU call L: TC part cap for shares; big graph test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
S="python scripts/sim_scale.py --reps 3"
for kk in 128 64 32 16; do echo "== tct part $kk K keys" ; GM_TCT_PART_KKEYS=$kk $S --workload tc --scale 22 --ef 10 --worlds 8; done 2>&1 | grep -v amdgpu.ids | tee $O/sim_tc_parts.txt
for kk in 1024 256 64; do echo "== tct part $kk K keys, world 1" ; GM_TCT_PART_KKEYS=$kk $S --workload tc --scale 22 --ef 10 --worlds 1; done 2>&1 | grep -v amdgpu.ids | tee -a $O/sim_tc_parts.txt
(time timeout 1700 python -m pytest tests/test_gpu_fullsize.py -q -x -k "2e31" -rs -s) 2>&1 | grep -v amdgpu.ids | tail -8 > $O/pytest_big.log; cat $O/pytest_big.log
