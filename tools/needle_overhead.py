import sys, json, numpy as np
sys.path.insert(0,'/root/repo')
import dxrvoxelizer_amd as dxv
from dxrvoxelizer_amd import meshes
vb, ib = meshes.uv_sphere(64, 32, 1.0)
vb[:,1] *= 0.004; vb[:,2] *= 0.004     # needle along x: almost every voxel is culled by the root early-out
v = dxv.Voxelizer(0); v.InitFromArrays(vb, ib)
for N in (256, 512, 1024):
    v.Voxelize(N); ts=[]
    for _ in range(7):
        v.Voxelize(N); ts.append(v.stats()['voxelize_ms'])
    print(json.dumps({'needle_N': N, 'ms': float(np.median(ts)), 'solid': v.CountSolid(), 'waves': N**3//64}))
