#!/usr/bin/env python3
"""Per-kernel PMC summary of one bench.py workload profiled by scripts/profile_configs.sh: counters per launch and the unit
utilisations they imply under the CALIBRATED issue costs of profiles/r03/issue_calibration.txt (gfx950 at 2.4 GHz under load):
  VALU   between 2.4 (add / logic / shift / mov) and 4.1 cycles (everything else, incl. any VALU with an SGPR operand) per wave64
         instruction and SIMD -- a range, SQ_INSTS_VALU does not tell the classes apart; 1024 SIMDs
  SALU   1.0 cycle per instruction and CU (one scalar unit per CU); 256 CUs
  LDS    SQ_LDS_IDX_ACTIVE = busy cycles of the LDS, summed over the CUs (2 per ds_read_b32, 4 per ds_read_b128 / ds_write_b32, + conflicts)
usage: summarize_pmc.py <dir with pmc_*/ and trace/> <kernel name substrings,comma separated>"""
import collections
import csv
import glob
import os
import sys

root, pats = sys.argv[1], sys.argv[2].split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if not any(p in k for p in pats):
            continue
        a = agg[k[:90]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
dur = {}
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if any(p in row["Name"] for p in pats):
            dur[row["Name"][:90]] = (float(row["AverageNs"]) / 1e6, int(row["Calls"]))
for k in sorted(agg, key=lambda k: -dur.get(k, (0, 0))[0]):
    c = {n: v[0] / max(v[1], 1) for n, v in agg[k].items()}
    ms, calls = dur.get(k, (0.0, 0))
    print(f"== {k}\n   rocprofv3 --kernel-trace --stats: average {ms:.3f} ms over {calls} launches")
    for n in sorted(c):
        print(f"   {n:24s} {c[n]:20.1f} per launch")
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0  # summed over the 8 XCDs
    if cyc > 0:
        v, s, lds = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_LDS_IDX_ACTIVE", 0)
        print(f"   -> kernel cycles {cyc:.3e} ({cyc / (ms * 1e-3) / 1e9 if ms else 0:.2f} GHz); VALU issue {v * 2.4 / 1024 / cyc:.2f} .. {min(v * 4.1 / 1024 / cyc, 9.99):.2f} of the SIMDs' cycles "
              f"(all fast .. all slow class); SALU {s * 1.0 / 256 / cyc:.2f} of the scalar units'; LDS {lds / 256 / cyc:.2f} of the LDS' "
              f"({c.get('SQ_LDS_BANK_CONFLICT', 0) / max(lds, 1):.2f} of that bank conflicts); waves waiting {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f} of their cycles")
    if "FETCH_SIZE" in c:
        fb, wb = c["FETCH_SIZE"] * 2048.0, c.get("WRITE_SIZE", 0) * 1024.0  # KB units; FETCH x2: gfx950 calibration
        print(f"   -> traffic past L2: fetch {fb / 1e9:.2f} GB + write {wb / 1e9:.2f} GB per launch = {(fb + wb) / (ms * 1e-3) / 1e12 if ms else 0:.2f} TB/s; "
              f"L2 hit rate {c.get('TCC_HIT_sum', 0) / max(c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0), 1):.2f}")
