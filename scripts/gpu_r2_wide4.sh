#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r2p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setops.py tests/test_gpu_cli.py -m gpu -x -q -k "clique or diamond_listing or count_smaller or sort_neighbors or unsorted or deeper or fails_loudly" 2>&1 | tail -4 | tee $O/pytest.log
C='clique4_rmat22:--workload;clique4;--steps;3;--warmup;1'
C2='clique4_rmat20:--workload;clique4;--scale;20;--ef;16;--steps;5;--warmup;1'
( for w in 2048 1024 512 256 128; do echo "== GM_WIDE_MIN_WORDS=$w"; GM_WIDE_MIN_WORDS=$w python scripts/ab.py $O/a_$w.json default "$C" "$C2"; done
  echo "== min words 512, tests"; GM_WIDE_MIN_WORDS=512 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "clique" 2>&1 | tail -2
) 2>&1 | tee $O/ab_wide4.log
