#!/bin/bash
# round 3, GPU call J: per-kernel durations of one rank's share (rank 0 of 8), diamond / motif3 / tc / clique4; why was the big test skipped
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $O
cat > /tmp/share.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from graphminer_amd import CliqueSolver, MotifSolver, SglSolver, TCSolver
from graphminer_amd.rmat import rmat_csr_device
w, sc, ef, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sym, _rp, _ci = rmat_csr_device(sc, ef, 42, 0)
dag = sym.orient()
run = {"tc": lambda **kw: TCSolver(dag, return_stats=True, **kw), "diamond": lambda **kw: SglSolver(sym, "diamond", return_stats=True, **kw),
       "clique4": lambda **kw: CliqueSolver(dag, 4, return_stats=True, **kw), "motif3": lambda **kw: MotifSolver(sym, 3, return_stats=True, **kw)}[w]
for i in range(6):
    _, st = run(rank=0, world=world)
print(w, world, "kernel_ms", st.kernel_ms, "chunks", st.chunks)
PY
cd /tmp
for spec in "diamond 22 10 8" "diamond 22 10 1" "motif3 24 16 8" "tc 22 10 8" "clique4 22 28 8"; do
  n=$(echo $spec | tr ' ' '_')
  rocprofv3 --kernel-trace --stats -d $O/p_$n -o t --output-format csv -- python /tmp/share.py $spec > $O/$n.log 2>&1
  f=$(find $O/p_$n -name "*kernel_stats.csv" | head -1); echo "== $spec"; tail -1 $O/$n.log; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'gm::' in r['Name']: print(f\"   {r['Name'][:70]:70s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms  max {float(r['MaxNs'])/1e6:8.3f}\")
"
  rm -rf $O/p_$n
done
cd $GRAFT_REPO_ROOT
