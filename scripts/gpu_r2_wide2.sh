#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
C='clique4_rmat22:--workload;clique4;--steps;3;--warmup;1'
( echo "== w8 batch 64 profile"; GM_WIDE_PROFILE=1 GM_WIDE_BATCH=64 python scripts/ab.py $O/a1.json default "$C"
  echo "== w8 batch 16 profile"; GM_WIDE_PROFILE=1 GM_WIDE_BATCH=16 python scripts/ab.py $O/a2.json default "$C"
  echo "== w8 batch 16"; GM_WIDE_BATCH=16 python scripts/ab.py $O/a3.json default "$C"
  echo "== w8 batch 8"; GM_WIDE_BATCH=8 python scripts/ab.py $O/a4.json default "$C"
  echo "== w12/w16 batch 16 profile"; GM_WIDE_PROFILE=1 GM_WIDE_BATCH=16 python scripts/ab.py $O/a5.json w12,w16 "$C"
  echo "== w12/w16 batch 16"; GM_WIDE_BATCH=16 python scripts/ab.py $O/a6.json w12,w16 "$C"
  echo "== w16 batch 8"; GM_WIDE_BATCH=8 python scripts/ab.py $O/a7.json w16 "$C"
) 2>&1 | tee $O/ab_wide2.log
