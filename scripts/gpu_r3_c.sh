#!/bin/bash
# round 3, GPU call C: long calibration runs; 4-clique with the topological renumbering + triangular counts; whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
python scripts/issue_calibration.py --pmc > $O/issue_calibration.txt 2> $O/issue_calibration.err; echo "calib rc=$?"
GM_WIDE_PROFILE=1 timeout 600 python bench.py --workload clique4 --steps 5 --warmup 2 --no-cpu-baseline --traffic off > $O/clique4.json 2> $O/clique4.err; echo "clique4 rc=$?"
GM_CLIQUE_NO_TOPO=1 timeout 600 python bench.py --workload clique4 --steps 5 --warmup 2 --no-cpu-baseline --traffic off > $O/clique4_notrim.json 2> $O/clique4_notrim.err; echo "clique4 notrim rc=$?"
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_clique4 -o c4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload clique4 --steps 3 --warmup 1 --no-cpu-baseline --traffic off > /dev/null 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof_clique4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/clique4_kernel_stats.csv; rm -rf $O/prof_clique4
python - <<'PY'
import json
for f in ("clique4","clique4_notrim"):
    try:
        d=json.load(open(f"gpurun_out/r3c/{f}.json")); print(f, d["kernel_ms_avg"], d["count"], d["setup_ms"], d["first_call_ms"], d["roofline"].get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
head -9 $O/clique4_kernel_stats.csv | cut -c1-130
grep "clique plan" $O/clique4.err | head -2
