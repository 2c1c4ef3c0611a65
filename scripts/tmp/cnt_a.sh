out=gpurun_out/r3cnt_a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "clique" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
export TMPDIR=/tmp
python bench.py --workload clique4 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/clique4.json 2> $out/clique4.err; echo "rc=$?"
rocprofv3 --kernel-trace --stats -d $out/prof -o clique4 -- python bench.py --workload clique4 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/prof.log 2>&1
python - <<'PY'
import json,glob,csv
d=json.loads(open('gpurun_out/r3cnt_a/clique4.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('count'), d.get('count_matches_cpu'))
for f in glob.glob('gpurun_out/r3cnt_a/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        if 'gm::' in r['Name']: print(r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e6)
PY
