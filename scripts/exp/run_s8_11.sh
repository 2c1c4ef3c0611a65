export TMPDIR=/tmp
for w in tc motif3; do
    GM_KST_PROBE=1 GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i "probe\|key stream" | head -10
done
