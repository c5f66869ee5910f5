#!/usr/bin/env python3
"""The lists' superset claim checked exhaustively on the device (dxv_debug_list_check) over BASELINE configs 2-5:
for every voxel, every triangle the canonical triangle step accepts for the ray must be selectable from the ray's texel
list.  One JSON line per configuration: accepted (ray, triangle) pairs, violations (must be 0).
usage: list_check_configs.py [--quick]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402


def main():
    quick = "--quick" in sys.argv
    v = dxv.Voxelizer(0)
    v.set_option("lists", 2)
    jobs = [("2: bunny 256^3", "bunny", 256), ("3: dragon x9 512^3", "dragon9", 512), ("metric: torus-1M 512^3", "torus1m", 512),
            ("metric: bunny x16 512^3", "bunny16", 512)]
    if not quick:
        jobs += [("4: dragon x9 1024^3", "dragon9", 1024), ("5: soup-10M 512^3", "soup10m", 512)]
    bad = 0
    for label, mesh, N in jobs:
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        t0 = time.perf_counter()
        accepted, violations, first = v.list_check(N)
        st = v.stats()
        print(json.dumps({"config": label, "N": N, "triangles": st["num_tris"], "accepted_ray_triangle_pairs": accepted,
                          "violations": violations, "first": first, "seconds": round(time.perf_counter() - t0, 2)}), flush=True)
        bad += violations
    print("VIOLATIONS", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
