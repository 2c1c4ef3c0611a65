#!/bin/bash
# round 2, final profiles: rocprofv3 kernel-trace stats for every BASELINE workload (+ diamond on R-MAT-24), PMC passes (SQ sets,
# FETCH_SIZE, WRITE_SIZE, separate runs) for TC R-MAT-22, diamond R-MAT-22 / R-MAT-24 and 3-motif R-MAT-24.  Summaries -> gpurun_out/$1/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$PWD
export TMPDIR=/tmp
O=$REPO/gpurun_out/${1:-prof_final}; mkdir -p $O
B="--no-cpu-baseline --traffic off --steps 5 --warmup 1"
declare -A CASES=( [tc_rmat22]="--workload tc" [diamond_rmat22]="--workload diamond" [clique4_rmat22ef28]="--workload clique4" [motif3_rmat24]="--workload motif3"
                   [diamond_rmat24]="--workload diamond --scale 24 --ef 16" [tc_uniform]="--workload tc --uniform 4847571,43000000" [tc_powerlaw]="--workload tc --powerlaw 4847571,43000000,20000" )
cd /tmp
for name in tc_rmat22 tc_uniform tc_powerlaw diamond_rmat22 clique4_rmat22ef28 motif3_rmat24 diamond_rmat24; do
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name/trace -o trace -- python $REPO/bench.py ${CASES[$name]} $B > $O/${name}_bench_line.json 2>/dev/null
  find /tmp/p_$name/trace -name "*kernel_stats.csv" -exec sh -c 'head -12 "$1" | cut -c1-200 > "$2"' _ {} $O/${name}_kernel_stats.csv \;
done
for name in tc_rmat22 tc_uniform diamond_rmat22 diamond_rmat24 motif3_rmat24 clique4_rmat22ef28; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
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
            if not any(x in k for x in ("mine_kernel", "hrow_kernel", "giant_kernel", "tct_kernel", "clique_build", "clique_count")) or "mine_kernel<6" in k: continue
            agg[(k[:58], row.get("Counter_Name"))][0] += float(row.get("Counter_Value", 0)); agg[(k[:58], row.get("Counter_Name"))][1] += 1
        for (k, c), (s, n) in sorted(agg.items()):
            print(f"{k:58s} {c:24s} per-launch {s/n:18.1f}  launches {n}")
PY
done
cd $REPO
ls $O | head -40
for name in diamond_rmat22 diamond_rmat24 motif3_rmat24 clique4_rmat22ef28; do echo "== $name"; head -7 $O/${name}_kernel_stats.csv | cut -c1-130; done
