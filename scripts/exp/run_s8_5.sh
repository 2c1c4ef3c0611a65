export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/s8/kt
for w in ${1:-motif3 clique4}; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o t --output-format csv -- python $R/bench.py --workload $w --steps 2 --warmup 0 --traffic off --no-cpu-baseline > /dev/null 2> /tmp/kt_$w.err)
  f=$(find /tmp/kt_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w $f"
  cp $f gpurun_out/s8/kt/${w}_kernel_stats.csv
  python - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
done
