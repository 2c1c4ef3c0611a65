#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python scripts/exp/row_range_stats.py 24 16
python scripts/exp/row_range_stats.py 22 10
