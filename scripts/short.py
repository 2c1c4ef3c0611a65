import json, sys
for line in sys.stdin:
    try:
        d = json.loads(line)
    except Exception:
        print(line.rstrip()); continue
    r = d.get("roofline") or {}
    print(f"{d['config']['graph']} {d['config']['workload'].split(':')[0]} n={d['n_gpus']} value={d['value']} {d['unit']} ms/step={d['ms_per_step']} "
          f"kernel_ms={d['kernel_ms_avg']} count={d['count']} roof={r.get('achieved')}GB/s frac={r.get('frac')}")
