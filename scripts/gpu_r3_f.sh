#!/bin/bash
# round 3, GPU call F: why is TC on the power-law stand-in slower on the topological view? + setup times with the temp pool
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
run() { n=$1; shift; timeout 600 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$?"; }
PL="--powerlaw 4847571,43000000,20000"
UN="--uniform 4847571,43000000"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --traffic off --workload tc"
GM_TABLE_INFO=1 run pl_topo_trim $B $PL
GM_TABLE_INFO=1 GM_TC_NO_TRIM=1 run pl_topo_notrim $B $PL
GM_TABLE_INFO=1 run pl_asnumbered $B $PL --tune 0,0,0,0,0,0,512
run un_topo_trim $B $UN
GM_TC_NO_TRIM=1 run un_topo_notrim $B $UN
run un_asnumbered $B $UN --tune 0,0,0,0,0,0,512
run r22_topo_trim $B
GM_TC_NO_TRIM=1 run r22_topo_notrim $B
run diamond python bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off --workload diamond
run motif3 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off --workload motif3
run clique4 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off --workload clique4
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["kernel_ms_avg"], d["count"], {k:round(v,1) for k,v in d["setup_ms"].items()}, round(d["first_call_ms"],1), d["roofline"].get("own_streamed_keys_per_launch"))
    except Exception as e: print(f, "ERR", e)
PY
grep -h "table" $O/pl_*.err | head -12
