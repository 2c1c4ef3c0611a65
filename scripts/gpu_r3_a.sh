#!/bin/bash
# round 3, GPU call A: issue-rate calibration (+PMC), the GPU test suite after the source split, the default bench line at round start
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
python scripts/issue_calibration.py --pmc > gpurun_out/r3a/issue_calibration.txt 2> gpurun_out/r3a/issue_calibration.err
echo "calib rc=$?"
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r3a/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest_gpu.log
tail -3 gpurun_out/r3a/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3a/bench_start.json 2> gpurun_out/r3a/bench_start.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/issue_calibration.err
