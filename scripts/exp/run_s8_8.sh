export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_u -o t --output-format csv -- python $R/bench.py --workload tc --uniform 4847571,43000000 --steps 2 --warmup 0 --traffic off --no-cpu-baseline > /dev/null 2> /tmp/kt_u.err)
f=$(find /tmp/kt_u -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:60]:
    n=r['Name']
    if n.startswith('void at::') or 'rocprim' in n and 'onesweep' in n: continue
    print(f"{n[:100]:100s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
GM_SETUP_TRACE=1 python bench.py --workload tc --uniform 4847571,43000000 --steps 3 --warmup 1 --traffic off --no-cpu-baseline 2>&1 >/dev/null | grep setup | head -30
