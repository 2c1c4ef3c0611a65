export TMPDIR=/tmp
mkdir -p gpurun_out/r04_end2
( time timeout 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q -x ) > gpurun_out/r04_end2/pytest_bench.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04_end2/pytest_bench.log
timeout 1200 python bench.py --steps 20 --warmup 3 > gpurun_out/r04_end2/bench_default_line.json 2> gpurun_out/r04_end2/bench_default.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r04_end2/bench_default_line.json
