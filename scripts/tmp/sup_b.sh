out=gpurun_out/r3sup_b; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -x -k "hashed_sets or diamond or clique or colliding" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
python bench.py --workload clique4 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/clique4.json 2> $out/clique4.err; echo "rc=$?"
python bench.py --workload diamond --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/diamond.json 2> $out/diamond.err; echo "rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3sup_b/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('kernel_ms_avg'), d.get('count'), d.get('count_matches_cpu'), d.get('setup_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
