export TMPDIR=/tmp
mkdir -p gpurun_out/s8
( time timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "renumbered or topological or planted" ) > gpurun_out/s8/pytest_sub2.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s8/pytest_sub2.log
for w in tc motif3 clique4; do
  for v in new; do
    GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --traffic off --no-cpu-baseline > gpurun_out/s8/trace2_${w}_$v.json 2> gpurun_out/s8/trace2_${w}_$v.err
    echo "== $w $v rc=$?"; grep -i "relabel" gpurun_out/s8/trace2_${w}_$v.err | head -12
    python - gpurun_out/s8/trace2_${w}_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d.get("count"), d.get("kernel_ms_avg"), d.get("setup_ms"), d.get("first_call_ms"))
PY
  done
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/s8/prof_relabel -o clique4 -- python /root/repo/bench.py --workload clique4 --steps 1 --warmup 0 --traffic off --no-cpu-baseline > /dev/null 2>&1
cd /root/repo; f=$(ls gpurun_out/s8/prof_relabel/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "relabel\|orient\|Name" $f | cut -c1-200
rm -rf gpurun_out/s8/prof_relabel/*.db gpurun_out/s8/prof_relabel/*trace.csv
