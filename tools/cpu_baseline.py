#!/usr/bin/env python3
"""CPU baseline table: the oracle's scalar BVH voxelizer,
both occupancy rules, one thread and all host cores, on the box it runs on.  One JSON line per row.
The reference has no CPU path of its own; this is the 'port' baseline bench.py also samples."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_mesh  # noqa: E402
from oracle import orc  # noqa: E402


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


cores = orc.usable_cores()
for mesh in ("bunny", "dragon", "torus1m"):
    vb, ib, label = make_mesh(mesh)
    t0 = time.perf_counter()
    scene = orc.Scene(vb, ib)
    build_s = time.perf_counter() - t0
    for N in (64, 256):
        for mode, tag in ((orc.MODE_REFERENCE, "reference"), (orc.MODE_PARITY, "parity")):
            for threads in (1, cores):
                if threads == 1 and N == 256:
                    zs = list(range(0, N, 8))               # bounded sample for the single thread
                else:
                    zs = list(range(N))
                dt = timed(lambda: orc.voxelize_slices(scene, N, zs, mode=mode, threads=threads))
                print(json.dumps({"mesh": mesh, "tris": len(ib) // 3, "N": N, "mode": tag, "threads": threads,
                                  "slices": len(zs), "seconds": dt, "mvox_s": len(zs) * N * N / dt / 1e6,
                                  "oracle_bvh_build_s": build_s}), flush=True)
