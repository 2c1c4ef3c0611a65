out=gpurun_out/r3sup_a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "diamond or scheduling or everything or planted or hub or sym" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
python bench.py --workload diamond --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/diamond.json 2> $out/diamond.err; echo "rc=$?"
GM_DIAMOND_PER_EDGE=1 python bench.py --workload diamond --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/diamond_pe.json 2> $out/diamond_pe.err; echo "rc=$?"
python bench.py --workload diamond --scale 24 --ef 16 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/diamond24.json 2> $out/diamond24.err; echo "rc=$?"
python bench.py --workload diamond --powerlaw 4847571,43000000,20000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/diamondp.json 2> $out/diamondp.err; echo "rc=$?"
GM_DIAMOND_PER_EDGE=1 python bench.py --workload diamond --powerlaw 4847571,43000000,20000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/diamondp_pe.json 2> $out/diamondp_pe.err; echo "rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3sup_a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('kernel_ms_avg'), d.get('count'), d.get('count_matches_cpu'), d.get('setup_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $out/*.err | tail -20
