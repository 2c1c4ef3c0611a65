export TMPDIR=/tmp
mkdir -p gpurun_out/s8
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "renumbered or orientation or topological" ) > gpurun_out/s8/pytest_sub3.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s8/pytest_sub3.log
for w in tc motif3 clique4; do
    GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --traffic off --no-cpu-baseline > gpurun_out/s8/trace3_${w}.json 2> gpurun_out/s8/trace3_${w}.err
    echo "== $w rc=$?"; grep -i "orient" gpurun_out/s8/trace3_${w}.err | head -6
    python - gpurun_out/s8/trace3_${w}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d.get("count"), d.get("kernel_ms_avg"), d.get("setup_ms"), d.get("first_call_ms"))
PY
done
