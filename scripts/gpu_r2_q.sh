#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
for w in 0 8 16 32; do
echo "GM_CLS_SWEEP=$w"
if [ $w = 0 ]; then unset GM_CLS_SWEEP; else export GM_CLS_SWEEP=$w; fi
python scripts/ab.py gpurun_out/r2s/a.json default 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1' 2>&1 | cut -c1-110
done | tee gpurun_out/r2s/cls_sweep.log
