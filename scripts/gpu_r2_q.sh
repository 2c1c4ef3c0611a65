#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
{
for cap in 32 8 2 1; do
echo "== GM_CLS_CAP_MKEYS=$cap, classes forced"
GM_CLS_CAP_MKEYS=$cap python scripts/sim_scale.py --workload diamond --scale 22 --ef 10 --reps 2 --tune 0,0,0,0,0,0,1048576,0 --worlds 1,2,8
GM_CLS_CAP_MKEYS=$cap python scripts/sim_scale.py --workload motif3 --scale 24 --ef 16 --reps 2 --tune 0,0,0,0,0,0,1048576,0 --worlds 1,8
GM_CLS_CAP_MKEYS=$cap python scripts/sim_scale.py --workload diamond --scale 20 --ef 16 --reps 2 --tune 0,0,0,0,0,0,1048576,0 --worlds 1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s/cls_cap.log
