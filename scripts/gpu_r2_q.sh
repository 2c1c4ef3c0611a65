#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
for lo in 3072 2048 1024 512 256; do
echo "GM_CLS_LO=$lo"
GM_CLS_LO=$lo python scripts/ab.py gpurun_out/r2s/a.json default 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1' 'diamond_rmat22:--workload;diamond;--steps;5;--warmup;1' 'diamond_rmat20:--workload;diamond;--scale;20;--ef;16;--steps;5;--warmup;1' 2>&1 | cut -c1-110
done | tee gpurun_out/r2s/hrow_lo2.log
