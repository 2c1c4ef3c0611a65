#!/bin/bash
# round 3, GPU call I: rank shares, round-2 library vs now on the same box; big-handle tests again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
R2=graphminer_amd/variants/r2tree
for w in diamond motif3; do
  sc=22; ef=10; [ $w = motif3 ] && sc=24 && ef=16
  echo "== $w r2" >> $O/sim_ab.txt; (cd $R2 && python scripts/sim_scale.py --reps 3 --workload $w --scale $sc --ef $ef --worlds 1,8) >> $O/sim_ab.txt 2>&1
  echo "== $w now" >> $O/sim_ab.txt; python scripts/sim_scale.py --reps 3 --workload $w --scale $sc --ef $ef --worlds 1,8 >> $O/sim_ab.txt 2>&1
done
echo "== tc r2" >> $O/sim_ab.txt; (cd $R2 && python scripts/sim_scale.py --reps 3 --workload tc --worlds 1,8) >> $O/sim_ab.txt 2>&1
grep -v amdgpu.ids $O/sim_ab.txt
(time timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "big_handle or 2e31" -s) > $O/pytest_big.log 2>&1; echo "pytest big rc=$?"; grep -v amdgpu.ids $O/pytest_big.log | tail -8
