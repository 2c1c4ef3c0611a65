#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "hub_paths or adversarial or motif or diamond or hub_graph or rows_longer" 2>&1 | tail -2
python scripts/ab.py gpurun_out/r2s/a.json default 'motif3_rmat24:--workload;motif3;--steps;3;--warmup;1' 'diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1' 'diamond_rmat22:--workload;diamond;--steps;5;--warmup;1' 2>&1 | cut -c1-110
