#!/bin/bash
# rocprofv3 kernel trace + PMC passes of bench.py (args: extra bench args). Only small CSV summaries are kept.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$PWD
export TMPDIR=/tmp
TAG=${PROF_TAG:-prof}
OUT=/tmp/gmprof_$TAG
KEEP=$REPO/gpurun_out/$TAG
rm -rf $OUT $KEEP; mkdir -p $OUT $KEEP
ARGS="${@:---scale 18 --ef 16 --steps 5 --warmup 1 --no-cpu-baseline}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $KEEP/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o pmc -- python $REPO/bench.py $ARGS > $KEEP/pmc_$i.log 2>&1
done
cd $REPO
find $OUT -name "*kernel_stats.csv" -exec cp {} $KEEP/kernel_stats.csv \;
python scripts/summarize_prof.py $OUT > $KEEP/summary.txt 2>&1
cat $KEEP/summary.txt | head -80
find $OUT -type f | head -30
tail -2 $KEEP/trace.log
