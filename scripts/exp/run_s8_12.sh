export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "renumbered or sgl or rectangle or house or pentagon or motif or hub" ) 2>&1 | tail -3
GM_SETUP_TRACE=1 timeout 600 python bench.py --workload motif3e --steps 2 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i "relabel" | head -12
GM_RELABEL_GLOBAL_SORT=1 GM_SETUP_TRACE=1 timeout 600 python bench.py --workload motif3e --steps 2 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i "relabel" | head -12
