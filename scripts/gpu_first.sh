#!/bin/bash
# first GPU contact: self test, parity tests, a small and the default bench
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
lscpu | head -20 > gpurun_out/lscpu.txt 2>&1; nproc >> gpurun_out/lscpu.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log | tail -30
timeout 300 python bench.py --scale 18 --ef 16 --steps 5 --warmup 1 --cpu-seconds 3 > gpurun_out/bench_s18.log 2>&1
tail -3 gpurun_out/bench_s18.log
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 10 > gpurun_out/bench_default.log 2>&1
tail -3 gpurun_out/bench_default.log
