#!/bin/bash
# round 3, GPU call G: the topological-view switch; tests that changed; default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
run() { n=$1; shift; timeout 600 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"; }
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --traffic off --workload"
GM_TABLE_INFO=1 run pl_tc $B tc --powerlaw 4847571,43000000,20000
GM_TABLE_INFO=1 run un_tc $B tc --uniform 4847571,43000000
GM_TABLE_INFO=1 run r22_tc $B tc
GM_TABLE_INFO=1 run r22ef28_clique4 $B clique4
GM_TABLE_INFO=1 run r24_motif3f $B motif3f
GM_TABLE_INFO=1 run r20_clique4 $B clique4 --scale 20 --ef 16
grep -h "topo view" $O/*.err | sort | uniq -c
(time timeout 1800 python -m pytest tests -m gpu -q) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["kernel_ms_avg"], d["count"], {k:round(v,1) for k,v in d["setup_ms"].items()}, round(d["first_call_ms"],1), d["roofline"].get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
