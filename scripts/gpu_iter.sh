#!/bin/bash
# iteration loop on the GPU box: parity tests, then bench lines (extra args = extra bench configs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "--- s18"; timeout 300 python bench.py --scale 18 --ef 16 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python scripts/short.py
echo "--- default"; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python scripts/short.py
for extra in "$@"; do echo "--- $extra"; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $extra 2>&1 | tail -1 | python scripts/short.py; done
