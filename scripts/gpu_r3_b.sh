#!/bin/bash
# round 3, GPU call B: calibration with measured residency; the re-hosted 4-clique build (tests + bench + A/B without the topological trim)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3b
mkdir -p $O
python scripts/issue_calibration.py > $O/issue_calibration.txt 2> $O/issue_calibration.err; echo "calib rc=$?"
GM_CAL_LDS=16384 python scripts/issue_calibration.py > $O/issue_calibration_lds16k.txt 2>> $O/issue_calibration.err; echo "calib16k rc=$?"
(time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "clique or planted or golden") > $O/pytest_clique.log 2>&1; echo "pytest clique rc=$?"; tail -3 $O/pytest_clique.log
GM_WIDE_PROFILE=1 timeout 600 python bench.py --workload clique4 --steps 5 --warmup 2 --no-cpu-baseline --traffic off > $O/clique4.json 2> $O/clique4.err; echo "clique4 rc=$?"
GM_CLIQUE_NO_TOPO=1 timeout 600 python bench.py --workload clique4 --steps 5 --warmup 2 --no-cpu-baseline --traffic off > $O/clique4_notopo.json 2> $O/clique4_notopo.err; echo "clique4 notopo rc=$?"
timeout 600 python bench.py --workload tc --steps 10 --warmup 2 --no-cpu-baseline --traffic off > $O/tc.json 2> $O/tc.err; echo "tc rc=$?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_clique4 -o c4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload clique4 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof_clique4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/clique4_kernel_stats.csv; rm -rf $O/prof_clique4
python - <<'PY'
import json
for f in ("clique4","clique4_notopo","tc"):
    try:
        d=json.load(open(f"gpurun_out/r3b/{f}.json")); print(f, d["kernel_ms_avg"], d["count"], d["setup_ms"], d["first_call_ms"])
    except Exception as e: print(f, "ERR", e)
PY
head -12 $O/clique4_kernel_stats.csv | cut -c1-150
grep "clique plan\|wide\]" $O/clique4.err | head
