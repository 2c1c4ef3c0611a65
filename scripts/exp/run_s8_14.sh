export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "renumbered or topological or planted" ) 2>&1 | tail -2
bash scripts/exp/run_s8_5.sh "motif3" | grep -i "relabel_rows"
