export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sorted or sort_neighbors or renumbered or orientation" ) 2>&1 | tail -3
bash scripts/exp/run_s8_5.sh "motif3" | grep -i "orient\|relabel\|sorted"
