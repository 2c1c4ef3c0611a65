#!/bin/bash
# round 3, GPU call E: TC on the topological view with trimmed in-edge tasks; the tests that changed
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
for w in "tc" "tc --uniform 4847571,43000000" "tc --powerlaw 4847571,43000000,20000" "motif3f" "clique4" "tc --scale 22 --ef 28"; do
  n=$(echo $w | tr ' ,' '__' | tr -d '-')
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --traffic off > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"
done
GM_TC_NOTOPO=1 timeout 600 python bench.py --workload tc --steps 10 --warmup 2 --no-cpu-baseline --traffic off --tune 0,0,0,0,0,0,512 > $O/tc_asnumbered.json 2> $O/tc_asnumbered.err
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_bench.py tests/test_gpu_dropin.py -q -x) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3e/*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["kernel_ms_avg"], d["count"], {k:round(v,1) for k,v in d["setup_ms"].items()}, round(d["first_call_ms"],1), d["roofline"].get("frac"), d["roofline"].get("own_streamed_keys_per_launch"))
    except Exception as e: print(f, "ERR", e)
PY
