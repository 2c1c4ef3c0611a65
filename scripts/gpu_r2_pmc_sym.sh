#!/bin/bash
# diamond / 3-motif on R-MAT-24: per-kernel time and SQ counters of the three workgroup classes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$PWD; export TMPDIR=/tmp
O=$REPO/gpurun_out/${1:-pmc_sym}; mkdir -p $O
B="--no-cpu-baseline --traffic off --steps 3 --warmup 1"
declare -A CASES=( [diamond_rmat24]="--workload diamond --scale 24 --ef 16" [motif3_rmat24]="--workload motif3" )
cd /tmp
for name in diamond_rmat24 motif3_rmat24; do
  rm -rf /tmp/p_$name
  GM_TABLE_INFO=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name/trace -o trace -- python $REPO/bench.py ${CASES[$name]} $B > $O/${name}_bench_line.json 2> $O/${name}_table_info.txt
  find /tmp/p_$name/trace -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p_$name/pmc_$i -o pmc -- python $REPO/bench.py ${CASES[$name]} $B > /dev/null 2>&1
  done
  python - /tmp/p_$name > $O/${name}_pmc_summary.txt <<'PY'
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if not any(x in k for x in ("mine_kernel", "hrow_kernel", "giant_kernel")): continue
            agg[(k[:58], row.get("Counter_Name"))][0] += float(row.get("Counter_Value", 0)); agg[(k[:58], row.get("Counter_Name"))][1] += 1
        for (k, c), (s, n) in sorted(agg.items()):
            print(f"{k:58s} {c:24s} per-launch {s/n:18.1f}  launches {n}")
PY
  head -6 $O/${name}_kernel_stats.csv | cut -c1-160
  grep "table/" $O/${name}_table_info.txt | cut -c1-220
  cat $O/${name}_pmc_summary.txt
done
