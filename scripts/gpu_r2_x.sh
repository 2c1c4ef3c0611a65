#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
python scripts/exp/dagdensity.py 22 28 2>&1 | tee gpurun_out/r2e/dagdensity_22_28.txt
