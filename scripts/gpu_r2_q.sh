#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
python scripts/ab.py gpurun_out/r2s/a.json default 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1' \
 'motif3_rmat22:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1' 'motif3_rmat22_cls:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,1048576,0' \
 'motif3_rmat18_cls:--workload;motif3;--scale;18;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,1048576,0' 2>&1 | tee gpurun_out/r2s/sym4.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
