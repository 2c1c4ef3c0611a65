export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tc_ or diamond or golden or short_rows or topological or support or key_stream or kst" ) 2>&1 | tail -3
for w in tc motif3; do
    GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i "key stream" | head -4
done
GM_SETUP_TRACE=1 timeout 600 python bench.py --workload tc --uniform 4847571,43000000 --steps 3 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i "key stream" | head -4
