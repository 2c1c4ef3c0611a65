#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hub_paths or adversarial" 2>&1 | tail -2
# a graph with nv = 2^25: classes (K = 25: the 32-bit multiply, class 1 at its limit) against the general path
python - <<'PY'
import sys; sys.path.insert(0, '.')
from graphminer_amd import SglSolver, MotifSolver
from graphminer_amd.rmat import rmat_csr_device
sym, rp, ci = rmat_csr_device(25, 4, 42, 0)
G = [0, 0, 0, 0, 0, 0, 0x80000]
d0, st0 = SglSolver(sym, "diamond", return_stats=True); d1, st1 = SglSolver(sym, "diamond", tune=G, return_stats=True)
m0, sm0 = MotifSolver(sym, 3, return_stats=True); m1, sm1 = MotifSolver(sym, 3, tune=G, return_stats=True)
print("R-MAT-25 ef4 diamond", d0, d0 == d1, "ms", round(st0.kernel_ms, 2), round(st1.kernel_ms, 2), "motif3", m0, m0 == m1, "ms", round(sm0.kernel_ms, 2), round(sm1.kernel_ms, 2))
PY
