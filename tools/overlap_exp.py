import sys, json, time, numpy as np
sys.path.insert(0, '.')
import dxrvoxelizer_amd as dxv
from bench import make_mesh
vb, ib, _ = make_mesh("torus1m")
ctxs = [dxv.Voxelizer(0) for _ in range(3)]
for v in ctxs: v.InitFromArrays(vb, ib)
def run(nctx, world, rank, steps=60):
    for v in ctxs[:nctx]:
        (v.VoxelizeInterleaved(512, rank, world, 8, 0) if world > 1 else v.Voxelize(512))
    t = time.perf_counter()
    for s in range(steps):
        v = ctxs[s % nctx]
        (v.VoxelizeInterleaved(512, rank, world, 8, 0, sync=False) if world > 1 else v.Voxelize(512, 0, 0, 512, sync=False))
    for v in ctxs[:nctx]: v.Sync()
    return (time.perf_counter() - t) / steps * 1e3
for world in (1, 8):
    for nctx in (1, 2, 3):
        print(json.dumps({"world": world, "contexts_in_flight": nctx, "ms_per_step": round(run(nctx, world, 3), 4)}))
