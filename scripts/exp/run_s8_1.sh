export TMPDIR=/tmp
mkdir -p gpurun_out/s8
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "renumbered or orientation or topological or planted or golden or tc_ or clique or diamond" ) > gpurun_out/s8/pytest_sub.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s8/pytest_sub.log
for w in tc motif3 clique4; do
  for v in new old; do
    if [ $v = old ]; then export GM_RELABEL_GLOBAL_SORT=1 GM_ORIENT_TWO_GATHERS=1; else unset GM_RELABEL_GLOBAL_SORT GM_ORIENT_TWO_GATHERS; fi
    GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --traffic off --no-cpu-baseline > gpurun_out/s8/trace_${w}_$v.json 2> gpurun_out/s8/trace_${w}_$v.err
    echo "== $w $v rc=$?"; grep -i "orient\|relabel" gpurun_out/s8/trace_${w}_$v.err | head -30
    python - gpurun_out/s8/trace_${w}_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d.get("count"), d.get("kernel_ms_avg"), d.get("setup_ms"), d.get("first_call_ms"))
PY
  done
done
