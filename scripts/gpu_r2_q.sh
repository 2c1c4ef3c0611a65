#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
{
python scripts/sim_scale.py --workload tc --scale 22 --ef 10 --reps 3
python scripts/sim_scale.py --workload diamond --scale 22 --ef 10 --reps 3
python scripts/sim_scale.py --workload clique4 --scale 22 --ef 28 --reps 2
python scripts/sim_scale.py --workload motif3 --scale 24 --ef 16 --reps 2
python scripts/sim_scale.py --workload diamond --scale 24 --ef 16 --reps 2 --worlds 1,8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2s/sim_scale_final.log
