export TMPDIR=/tmp
for pc in default 4 5 8 9 10 12 16 20; do
  if [ $pc = default ]; then unset GM_KST_WG_PER_CU; else export GM_KST_WG_PER_CU=$pc; fi
  for w in tc motif3; do
    echo "== per_cu=$pc $w: $(GM_SETUP_TRACE=1 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep -i 'key stream' | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4)}' | tr '\n' ' ')"
  done
done
