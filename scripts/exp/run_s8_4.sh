export TMPDIR=/tmp
mkdir -p gpurun_out/s8
tag=${1:-4}
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "${2:-renumbered or orientation or topological or tc_ or diamond or golden or short_rows or planted}" ) > gpurun_out/s8/pytest_sub$tag.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s8/pytest_sub$tag.log
for w in tc motif3 clique4 diamond; do
    GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --traffic off --no-cpu-baseline > gpurun_out/s8/trace${tag}_${w}.json 2> gpurun_out/s8/trace${tag}_${w}.err
    echo "== $w rc=$?"; grep -i "orient: seg\|orient: comp\|key stream\|edge desc\|task lists\|clique:\|table: begin" gpurun_out/s8/trace${tag}_${w}.err | head -14
    python - gpurun_out/s8/trace${tag}_${w}.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d.get("count"), d.get("kernel_ms_avg"), d.get("setup_ms"), d.get("first_call_ms"))
PY
done
