#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
for m in 0 1 2; do
echo "GM_CLIQUE_SIDE=$m"
GM_CLIQUE_SIDE=$m python scripts/ab.py gpurun_out/r2s/a.json default 'clique4:--workload;clique4;--steps;5;--warmup;1' 2>&1 | cut -c1-110
done | tee gpurun_out/r2s/clique_side.log
