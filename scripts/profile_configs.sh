#!/bin/bash
# rocprofv3 evidence for the BASELINE configs (run on the GPU box through gpurun): per-kernel durations (--kernel-trace --stats) and PMC
# counters in SEPARATE passes (never together with other trace domains), summarised with the calibrated issue costs into
# gpurun_out/prof/<name>_kernel_stats.csv and <name>_pmc_summary.txt -- copy what is to be judged into profiles/rNN/.
# usage: profile_configs.sh <out dir under gpurun_out> [config ...]   config = name:kernel-substrings:bench args (';' for spaces)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}; shift
mkdir -p $O
CFGS=("$@")
if [ ${#CFGS[@]} -eq 0 ]; then
  CFGS=("tc_rmat22:tch_kernel,core_tc_,mine_kernel<0:--workload;tc" "diamond_rmat22:sup_kernel<,core_tc_,sup_pairs_kernel,sup_near_kernel,sup_far_kernel:--workload;diamond"
        "clique4_rmat22ef28:cbuild_kernel,cgather_kernel,cgatherb_kernel,clique_mma_kernel,clique_small_kernel,mine_kernel<3:--workload;clique4"
        "motif3_rmat24:tch_kernel,core_tc_,mine_kernel<0:--workload;motif3" "motif3e_rmat24:hrow_kernel<2,giant_kernel<2,mine_kernel<2:--workload;motif3e" "diamond_rmat24:sup_kernel<,core_tc_,sup_pairs_kernel,sup_near_kernel,sup_far_kernel:--workload;diamond;--scale;24;--ef;16" "tc_uniform:tch_kernel:--workload;tc;--uniform;4847571,43000000"
        "tc_powerlaw:tch_kernel:--workload;tc;--powerlaw;4847571,43000000,20000")
fi
PMCG=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES"
        "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE")
for cfg in "${CFGS[@]}"; do
  name=${cfg%%:*}; rest=${cfg#*:}; kern=${rest%%:*}; args=$(echo ${rest#*:} | tr ';' ' ')
  W=$O/work_$name; rm -rf $W; mkdir -p $W
  B="python $GRAFT_REPO_ROOT/bench.py $args --steps 5 --warmup 1 --no-cpu-baseline --traffic off --first-call-repeats 0"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $W/trace -o t --output-format csv -- $B > $O/${name}_bench_line.json 2> $W/trace.err)
  i=0
  for grp in "${PMCG[@]}"; do
    (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $W/pmc_$i -o p --output-format csv -- $B > /dev/null 2> $W/pmc_$i.err)
    i=$((i+1))
  done
  find $W/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv
  python scripts/summarize_pmc.py $W "$kern" > $O/${name}_pmc_summary.txt 2>&1
  rm -rf $W
  echo "== $name"; head -c 300 $O/${name}_bench_line.json | tr ',' '\n' | grep -E "kernel_ms_avg|\"value\"" ; grep -E "^==|->" $O/${name}_pmc_summary.txt | cut -c1-260
done
