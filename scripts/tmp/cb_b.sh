out=gpurun_out/r3cb_b; mkdir -p $out
for w in 4 6 8; do
  GM_CB_WAVES=$w python bench.py --workload clique4 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/clique4_w$w.json 2> $out/clique4_w$w.err
  python - $out/clique4_w$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d.get('count_matches_cpu'))
PY
done
