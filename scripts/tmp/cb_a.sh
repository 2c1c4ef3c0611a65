out=gpurun_out/r3cb_a; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "clique or colliding" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
python bench.py --workload clique4 --steps 5 --warmup 1 --traffic off --no-cpu-baseline > $out/clique4.json 2> $out/clique4.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3cb_a/clique4.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('count'), d.get('count_matches_cpu'), d.get('setup_ms'))
PY
bash scripts/profile_configs.sh r3cb_a/prof "clique4_rmat22ef28:cbuild_kernel,clique_count_kernel,clique_small_kernel,mine_kernel<3:--workload;clique4"
