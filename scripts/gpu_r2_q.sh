#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r2s
C=1048576; N=524288
python scripts/ab.py gpurun_out/r2s/a.json default \
 "d_r18_gen:--workload;diamond;--scale;18;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "d_r18_cls:--workload;diamond;--scale;18;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_r20e10_gen:--workload;diamond;--scale;20;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "d_r20e10_cls:--workload;diamond;--scale;20;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "d_pl_gen:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "d_pl_cls:--workload;diamond;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r20_gen:--workload;motif3;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "m_r20_cls:--workload;motif3;--scale;20;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r22_gen:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "m_r22_cls:--workload;motif3;--scale;22;--ef;10;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r22e16_gen:--workload;motif3;--scale;22;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "m_r22e16_cls:--workload;motif3;--scale;22;--ef;16;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_r23_gen:--workload;motif3;--scale;23;--ef;16;--steps;3;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "m_r23_cls:--workload;motif3;--scale;23;--ef;16;--steps;3;--warmup;1;--tune;0,0,0,0,0,0,$C,0" \
 "m_pl_gen:--workload;motif3;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$N,0" "m_pl_cls:--workload;motif3;--powerlaw;4847571,43000000,20000;--steps;5;--warmup;1;--tune;0,0,0,0,0,0,$C,0" 2>&1 | cut -c1-100 | tee gpurun_out/r2s/cls_threshold2.log
