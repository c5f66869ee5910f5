"""Per-call cost at the reference's own grid size (64^3, bunny): kernel time from HIP events vs wall
time of synchronous and of back-to-back asynchronous calls."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

v = dxv.Voxelizer(0)
vb, ib, _ = make_mesh("bunny")
v.InitFromArrays(vb, ib)
for N in (64, 128, 256):
    for mode, name in ((dxv.MODE_REFERENCE, "reference"), (dxv.MODE_PARITY, "parity")):
        for _ in range(20):
            v.Voxelize(N, mode)
        t = time.perf_counter()
        for _ in range(200):
            v.Voxelize(N, mode)
        sync_us = (time.perf_counter() - t) / 200 * 1e6
        kern = v.stats()["voxelize_ms"] * 1e3
        t = time.perf_counter()
        for _ in range(200):
            v.Voxelize(N, mode, sync=False)
        v.Sync()
        async_us = (time.perf_counter() - t) / 200 * 1e6
        print(json.dumps({"mesh": "bunny", "N": N, "mode": name, "events_us": round(kern, 1), "sync_call_us": round(sync_us, 1),
                          "async_call_us": round(async_us, 1)}))
