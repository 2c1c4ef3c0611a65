#!/bin/bash
# HBM-traffic evidence for the bench kernel: FETCH_SIZE / WRITE_SIZE per launch + the dword-stream calibration
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
REPO=$PWD; export TMPDIR=/tmp
OUT=/tmp/gmtraffic; KEEP=$REPO/gpurun_out/traffic; rm -rf $OUT $KEEP; mkdir -p $OUT $KEEP
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib -o calib -- python $REPO/scripts/calib_fetch.py > $KEEP/calib.log 2>&1
for w in tc diamond clique4 motif3; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$w -o pmc -- python $REPO/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $KEEP/fetch_$w.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w_$w -o pmc -- python $REPO/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $KEEP/write_$w.log 2>&1
done
cd $REPO
python - <<'PY' > $KEEP/traffic_summary.txt
import csv, glob, collections
def agg(path, pat):
    a = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                k = (r["Kernel_Name"][:50], r["Counter_Name"]); a[k][0] += float(r["Counter_Value"]); a[k][1] += 1
    return {k: (s / n, n) for k, (s, n) in a.items()}
for k, v in agg("/tmp/gmtraffic/calib", "calib_stream").items(): print("CALIB", k, "per-launch", v)
for w in ("tc", "diamond", "clique4", "motif3"):
    for d in ("f_", "w_"):
        for k, v in agg("/tmp/gmtraffic/" + d + w, "mine_kernel").items(): print(w, k, "per-launch", v)
PY
cat $KEEP/traffic_summary.txt; tail -2 $KEEP/calib.log
