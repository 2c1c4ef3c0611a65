out=gpurun_out/r3sup_d; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_bench.py -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
