#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adversarial or hub_paths or hub_graph or random_graphs" --durations=5 2>&1 | tail -12
