#!/usr/bin/env python3
"""Diagnostic (library built with -DDXV_QUEUE_TIMES: python -c "from dxrvoxelizer_amd import build; build.build(defines=['DXV_QUEUE_TIMES'], name='qtimes')"):
when do the persistent waves of a queue launch start and end?  usage: DXV_LIBRARY=.../libdxv_qtimes.so queue_times.py [mesh] [grid] [opts]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
v.set_option("dispatch", 0)                               # (the stamps are the persistent waves')
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib)
share = int(os.environ.get("QT_WORLD", "1"))          # QT_WORLD=8: rank 0's share of the block-cyclic partition instead of the whole grid
for _ in range(3):
    v.VoxelizeInterleaved(N, 0, share, 4, 0) if share > 1 else v.Voxelize(N, 0)
st = v.stats()
raw = np.zeros(1 << 21, np.uint64)
v._check(v._lib.dxv_debug_download(v._ctx, 100, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
w = st["plan_waves"]
t = raw[:4 * w].reshape(w, 4).astype(np.int64)
t0 = t[:, 0].min()
start, end, last = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 3] - t0) / 100.0          # microseconds
bricks, longest = t[:, 2] >> 32, (t[:, 2] & 0xffffffff) / 100.0
q = [0, 1, 5, 25, 50, 75, 95, 99, 100]
pct = lambda a: dict(zip(q, np.percentile(a, q).round(1).tolist()))                                 # noqa: E731
started = start < 50.0                                      # (waves that were resident from the beginning)
print(json.dumps({"mesh": mesh, "N": N, "world": share, "resident_waves": int(started.sum()),
                  "busy_frac_of_resident": round(float((end[started] - start[started]).sum() / (end.max() * started.sum())), 3), "kernel_ms": round(st["voxelize_ms"], 4), "bricks": st["plan_bricks"], "waves": w,
                  "start_us_pct": pct(start), "end_us_pct": pct(end), "bricks_per_wave_pct": pct(bricks),
                  "longest_brick_us_pct": pct(longest), "last_brick_us_pct": pct(end - last),
                  "mean_brick_us": round(float((end - start).sum() / max(bricks.sum(), 1)), 2),
                  "idle_wave_us_at_end_mean": round(float((end.max() - end).mean()), 1),
                  "end_by_xcd_us": [round(float(end[x::8].max()), 1) for x in range(8)]}))
