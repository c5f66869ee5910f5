#!/usr/bin/env python3
"""Minimal driver for profilers (no torch): build the scene once, voxelize K times.
    python3 tools/run_once.py [mesh] [N] [K] [mode] [key=value ...] [world=W rank=R zblock=B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mode = dxv.MODE_PARITY if len(sys.argv) > 4 and sys.argv[4] == "parity" else dxv.MODE_REFERENCE
v = dxv.Voxelizer(0)
part = {"world": 1, "rank": 0, "zblock": 8, "prepare": 1}          # world=8 rank=0: one rank's share of the block-cyclic partition (bench.py --gpus 8);
                                                                    # prepare=0: Init is not told the grid (every launch builds its queue)
for kv in sys.argv[5:]:
    k, val = kv.split("=")
    if k in part:
        part[k] = int(val)
    else:
        v.set_option(k, int(val))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib)
if part["prepare"] and mode == dxv.MODE_REFERENCE:
    if part["world"] > 1:
        v.PrepareLaunchInterleaved(N, part["rank"], part["world"], part["zblock"])
    else:
        v.PrepareLaunch(N)
for _ in range(K):
    if part["world"] > 1:
        v.VoxelizeInterleaved(N, part["rank"], part["world"], part["zblock"], mode)
    else:
        v.Voxelize(N, mode)
print(v.stats(), v.CountSolid())
v.close()
