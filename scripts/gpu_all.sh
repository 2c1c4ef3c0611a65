#!/bin/bash
# one bench line per BASELINE.json config (single GPU)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for w in tc diamond clique4 motif3; do
  echo "--- $w"; timeout 900 python bench.py --workload $w --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python scripts/short.py
done
