#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
for a in "rmat 18 16" "rmat 20 16" "rmat 20 10" "rmat 22 10" "rmat 22 16" "rmat 23 16" "powerlaw"; do python scripts/exp/class_rows.py $a 2>/dev/null; done
C=1048576
python scripts/ab.py gpurun_out/r2s/a.json default \
 "d_r18:--workload;diamond;--scale;18;--ef;16;--steps;5;--warmup;1" "d_r18_cls:--workload;diamond;--scale;18;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_r20:--workload;diamond;--scale;20;--ef;16;--steps;5;--warmup;1" "d_r20_cls:--workload;diamond;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_r22e16:--workload;diamond;--scale;22;--ef;16;--steps;5;--warmup;1" "d_r22e16_cls:--workload;diamond;--scale;22;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_r23:--workload;diamond;--scale;23;--ef;16;--steps;3;--warmup;1" "d_r23_cls:--workload;diamond;--scale;23;--ef;16;--steps;3;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r20:--workload;motif3;--scale;20;--ef;16;--steps;5;--warmup;1" "m_r20_cls:--workload;motif3;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r22:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1" "m_r22_cls:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_pl:--workload;motif3;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1" "m_pl_cls:--workload;motif3;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_pl:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1" "d_pl_cls:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" 2>&1 | cut -c1-100 | tee gpurun_out/r2s/cls_threshold.log
