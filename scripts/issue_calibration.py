#!/usr/bin/env python3
"""Issue-rate calibration on the GPU box: what does ONE wave-instruction of each kind cost on gfx950, and what do
the SQ counters report for a unit that is known to be saturated?

  python scripts/issue_calibration.py                    in-kernel s_memtime table (gm_issue_calib), all kinds x 1/2/4/8 waves per SIMD
  python scripts/issue_calibration.py --pmc              + the same launches under rocprofv3 --pmc (separate passes per counter group)

Writes plain text to stdout; the GPU runner tees it into profiles/r03/issue_calibration.txt."""
import ctypes as C
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KINDS = ["v_add_u32", "v_mul_lo_u32", "v_mul_u32_u24", "v_cmp_lt_u32 -> sgpr pair", "v_add_u32 dpp row_shr:1", "s_add_u32", "ds_read_b128",
         "ds_read_b32", "ds_read_b32 32-way conflict", "v_add_u32 + s_add_u32 interleaved", "v_readlane_b32", "v_mbcnt_lo", "v_bcnt_u32_b32",
         "v_cmp_eq_u16 sdwa -> sgpr pair", "ds_write_b32", "v_cndmask_b32 (vcc)", "v_cndmask_b32_e64 (sgpr pair)", "v_and_b32", "v_ashrrev_i32",
         "v_min_u32", "v_mad_u32_u24", "v_add3_u32", "v_cmp_lt_u32 -> vcc", "v_cmp -> vcc + v_cndmask (32 + 32)", "v_mov_b32",
         "v_sub + v_ashr + v_and + v_add (16 x 4)", "v_xor_b32 with an sgpr source", "s_bcnt1_i32_b64"]
ITERS = 12000  # x 64 instructions per wave: milliseconds per launch, so that the dispatch ramp (~1 ms for 2048 workgroups) does not set the residency


def table(kinds, waves):
    import torch

    from graphminer_amd import _lib

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    lib = _lib.load()
    rows = []
    for k in kinds:
        for w in waves:
            a, b, ms, res = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            _lib.check(lib.gm_issue_calib(k, w, ITERS, C.byref(a), C.byref(b), C.byref(ms), C.byref(res)), "gm_issue_calib")
            rows.append((k, w, a.value, b.value, ms.value, res.value))
    return rows


def main():
    if "--worker" in sys.argv:
        kinds = [int(x) for x in sys.argv[sys.argv.index("--worker") + 1].split(",")]
        rows = table(kinds, [8])
        print("CAL_WORKER " + json.dumps(rows), flush=True)
        return
    rows = table(range(len(KINDS)), [1, 2, 4, 8])
    print("# gm_issue_calib: iters x 64 instructions of one kind per wave, W workgroups of 4 waves per CU requested, s_memtime around the loop")
    print("# cyc/inst = shader cycles per wave-instruction as ONE wave sees it (median over all waves); resid = waves of the launch in flight")
    print("# per SIMD at the same time, MEASURED from the per-wave start / end stamps and HW_ID; inst/cyc/SIMD = resid / (cyc/inst)")
    print(f"# {'kind':38s} {'W':>2s} {'cyc/inst(wave)':>15s} {'resid':>6s} {'inst/cyc/SIMD':>14s} {'inst/cyc/CU':>12s} {'ms':>8s}")
    for k, w, a, b, ms, res in rows:
        print(f"  {KINDS[k]:38s} {w:2d} {a:15.3f} {res:6.2f} {b:14.4f} {4 * b:12.4f} {ms:8.3f}")
    sat = {k: min(a / max(res, 1e-9) for kk, w, a, b, ms, res in rows if kk == k) for k in range(len(KINDS))}
    print("\n# saturated cost (min over W of cyc/inst / resid) = cycles of the unit per wave64 instruction:")
    for k in range(len(KINDS)):
        print(f"  {KINDS[k]:38s} {sat[k]:7.3f} cycles per SIMD" + (f"  = {sat[k] / 4:6.3f} per CU (one scalar unit)" if k == 5 else ""))
    if "--pmc" not in sys.argv:
        return
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    groups = [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES"], ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_SCA", "SQ_WAVES"],
              ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"], ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM"]]
    kinds = [0, 1, 3, 4, 5, 6, 7, 8, 9, 13, 15, 16, 23]
    print("\n# the same launches (W = 8, last of 3 repetitions counted 3x: sums over the 3 launches / 3) under rocprofv3 --pmc")
    got = {}
    for grp in groups:
        tmp = tempfile.mkdtemp(prefix="gm_cal_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", *grp, "--output-format", "csv", "-d", tmp, "-o", "cal", "--", sys.executable, os.path.abspath(__file__),
                   "--worker", ",".join(map(str, kinds))]
            r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=900)
            if r.returncode != 0:
                print(f"# rocprofv3 --pmc {' '.join(grp)} failed: {(r.stderr or r.stdout)[-300:]}")
                continue
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "issue_calib_kernel" not in name:
                        continue
                    kind = int(name.split("<")[1].split(">")[0])
                    d = got.setdefault(kind, {})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]) / 3.0
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    names = [c for g in groups for c in g]
    for k in kinds:
        if k not in got:
            continue
        print(f"  kind {k} {KINDS[k]}  (issued per launch: {64 * ITERS} per wave x 8192 waves = {64 * ITERS * 8192:.3e} wave-instructions)")
        for c in names:
            if c in got[k]:
                print(f"      {c:24s} {got[k][c]:16.0f}")
    print(json.dumps({"iters": ITERS, "rows": rows, "pmc": got}))


if __name__ == "__main__":
    main()
