#!/bin/bash
# round 3, GPU call H: big handles (incl. R-MAT-26 ef 20, > 2^31 entries); one-GPU simulation of the rank shares, class part caps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
(time timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "big_handle or 2e31 or topological_view" -s) > $O/pytest_big.log 2>&1; echo "pytest big rc=$?"; tail -6 $O/pytest_big.log
S="python scripts/sim_scale.py --reps 3"
for kk in 4096 1024 512 256; do
  echo "== class part cap $kk K keys, world 8" >> $O/sim_caps.txt
  GM_CLS_CAP_KKEYS=$kk $S --workload diamond --scale 22 --ef 10 --worlds 8 >> $O/sim_caps.txt 2>&1
  GM_CLS_CAP_KKEYS=$kk $S --workload motif3 --scale 24 --ef 16 --worlds 8 >> $O/sim_caps.txt 2>&1
done
for kk in 8192 2048 1024; do
  echo "== class part cap $kk K keys, world 4" >> $O/sim_caps.txt
  GM_CLS_CAP_KKEYS=$kk $S --workload diamond --scale 22 --ef 10 --worlds 4 >> $O/sim_caps.txt 2>&1
done
cat $O/sim_caps.txt
$S --workload tc --scale 22 --ef 10 > $O/sim_tc.txt 2>&1; cat $O/sim_tc.txt
$S --workload clique4 --scale 22 --ef 28 --worlds 1,8 > $O/sim_clique4.txt 2>&1; cat $O/sim_clique4.txt
