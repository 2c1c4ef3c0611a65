#!/bin/bash
# round 3, GPU call M: adaptive batches in tct / cbuild: shares and whole graphs; dequeue grab on the flat graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
S="python scripts/sim_scale.py --reps 3"
for kk in 128 64 32; do echo "== tct part $kk K keys" ; GM_TCT_PART_KKEYS=$kk $S --workload tc --scale 22 --ef 10 --worlds 1,8; done 2>&1 | grep -v amdgpu.ids | tee $O/sim_tc_parts.txt
$S --workload clique4 --scale 22 --ef 28 --worlds 1,8 2>&1 | grep -v amdgpu.ids | tee $O/sim_clique4.txt
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --traffic off --workload tc"
for gr in 0 2 4 8; do $B --uniform 4847571,43000000 --tune 0,$gr > $O/un_grab$gr.json 2>/dev/null; python -c "import json; d=json.load(open('$O/un_grab$gr.json')); print('uniform grab $gr', d['kernel_ms_avg'], d['count'])"; done
for gr in 0 2 4; do $B --powerlaw 4847571,43000000,20000 --tune 0,$gr > $O/pl_grab$gr.json 2>/dev/null; python -c "import json; d=json.load(open('$O/pl_grab$gr.json')); print('powerlaw grab $gr', d['kernel_ms_avg'], d['count'])"; done
for gr in 0 2; do $B --tune 0,$gr > $O/r22_grab$gr.json 2>/dev/null; python -c "import json; d=json.load(open('$O/r22_grab$gr.json')); print('rmat22 grab $gr', d['kernel_ms_avg'], d['count'])"; done
