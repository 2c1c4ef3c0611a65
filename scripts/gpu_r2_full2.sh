#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/gpu_r2_full.sh r2n
bash scripts/gpu_prof2.sh r2n_prof
