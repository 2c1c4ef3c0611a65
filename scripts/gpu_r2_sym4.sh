#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2o; mkdir -p $O
M='motif3_rmat24:--workload;motif3;--steps;3;--warmup;1'
D='diamond_rmat24:--workload;diamond;--scale;24;--ef;16;--steps;3;--warmup;1'
( echo "== default (one stream, 8M)"; python scripts/ab.py $O/a0.json default "$M" "$D"
  echo "== streams, 8M"; GM_CLASSES_STREAMS=1 python scripts/ab.py $O/a1.json default "$M" "$D"
  echo "== one stream, 32M"; GM_CLS_CAP_MKEYS=32 python scripts/ab.py $O/a2.json default "$M" "$D"
  echo "== streams, 32M"; GM_CLASSES_STREAMS=1 GM_CLS_CAP_MKEYS=32 python scripts/ab.py $O/a3.json default "$M" "$D"
  echo "== streams, 2M"; GM_CLASSES_STREAMS=1 GM_CLS_CAP_MKEYS=2 python scripts/ab.py $O/a4.json default "$M"
) 2>&1 | tee $O/ab_sym4.log
