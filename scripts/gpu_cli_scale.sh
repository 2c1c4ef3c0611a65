#!/bin/bash
# C++ CLI at scale: R-MAT-20 written in the three-file format, every binary against the Python/ctypes path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from graphminer_amd.rmat import rmat_csr_device
from graphminer_amd import TCSolver, SglSolver, CliqueSolver, MotifSolver
s, rp, ci = rmat_csr_device(20, 16, 42, 0)
g = s.download(); g.save("/tmp/rmat20/graph")
d = s.orient()
print("PY", TCSolver(d), SglSolver(s, "diamond"), CliqueSolver(d, 4), CliqueSolver(d, 5), MotifSolver(s, 3), flush=True)
PY
B=graphminer_amd/bin
$B/tc_gpu_base /tmp/rmat20/graph | tail -3
$B/sgl_gpu_base /tmp/rmat20/graph diamond | tail -2
$B/clique_gpu_base /tmp/rmat20/graph 4 | tail -2
$B/clique_gpu_base /tmp/rmat20/graph 5 | tail -1
$B/motif_gpu_base /tmp/rmat20/graph 3 | tail -3
GM_FORCE_RCCL_PATH=1 $B/tc_multigpu /tmp/rmat20/graph 1 | tail -4
