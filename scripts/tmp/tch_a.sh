out=gpurun_out/r3tch_a; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tc_ or triangle or topological or rows_beyond" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for w in tc; do
  python bench.py --workload $w --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/$w.json 2> $out/$w.err; echo "$w rc=$?"
  GM_TC_SORTED=1 python bench.py --workload $w --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/${w}_sorted.json 2> $out/${w}_sorted.err
done
python bench.py --workload tc --uniform 4847571,43000000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/tcu.json 2>$out/tcu.err
GM_TC_SORTED=1 python bench.py --workload tc --uniform 4847571,43000000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/tcu_sorted.json 2>$out/tcu_sorted.err
python bench.py --workload tc --powerlaw 4847571,43000000,20000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/tcp.json 2>$out/tcp.err
GM_TC_SORTED=1 python bench.py --workload tc --powerlaw 4847571,43000000,20000 --steps 10 --warmup 2 --traffic off --no-cpu-baseline > $out/tcp_sorted.json 2>$out/tcp_sorted.err
python bench.py --workload motif3f --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/m3f.json 2>$out/m3f.err
GM_TC_SORTED=1 python bench.py --workload motif3f --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/m3f_sorted.json 2>$out/m3f_sorted.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3tch_a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['ms_per_step'], d.get('kernel_ms_avg'), d.get('count'), d.get('count_matches_cpu'))
    except Exception as e: print(f, 'ERR', e)
PY
