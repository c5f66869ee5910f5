#!/usr/bin/env python3
"""dxv_debug_division_check over ALL even grid sizes up to 2048: the ray set-up's scale-free divisions (csrc/dxv_math.h: div_by) against IEEE
quotients for every voxel origin -- 2.2 x 10^12 origins, 14 words each.  One JSON line per block of sizes and a total; needs no mesh."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402

v = dxv.Voxelizer(0)
total = bad = 0
t00 = time.time()
for lo in range(2, 2049, 128):
    hi = min(lo + 126, 2048)
    t0 = time.time()
    checked, differing, first = v.division_check(lo, hi)
    total += checked
    bad += differing
    print(json.dumps({"grids": [lo, hi], "origins": checked, "differing": differing, "first": first, "seconds": round(time.time() - t0, 2)}), flush=True)
print(json.dumps({"all_even_grids_up_to": 2048, "origins": total, "expected": 8 * (1024 * 1025 // 2) ** 2, "differing": bad, "seconds": round(time.time() - t00, 1)}))
sys.exit(1 if bad or total != 8 * (1024 * 1025 // 2) ** 2 else 0)
