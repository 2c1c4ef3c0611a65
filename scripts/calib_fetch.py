#!/usr/bin/env python3
"""Run the dword-per-lane calibration stream over a 4 GiB buffer (>> 256 MiB Infinity Cache) 3 times.
Use under: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -o calib -- python scripts/calib_fetch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphminer_amd import _lib
lib = _lib.load()
n = 1 << 30
buf = torch.ones(n, dtype=torch.int32, device="cuda")
out = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(3):
    _lib.check(lib.gm_calib_stream(buf.data_ptr(), n, out.data_ptr(), None), "calib")
torch.cuda.synchronize()
print("bytes_per_launch", 4 * n, "sum", int(out.item()))
