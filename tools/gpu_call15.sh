#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu15.log 2>&1
python - > $OUT/render15.log 2>&1 <<'PY'
import sys, json, struct, numpy as np
sys.path.insert(0, '.')
import dxrvoxelizer_amd as dxv
from dxrvoxelizer_amd import camera
from bench import make_mesh
v = dxv.Voxelizer(0)
for mesh, N in (("bunny", 64), ("bunny", 256), ("dragon", 512)):
    vb, ib, _ = make_mesh(mesh)
    v.InitFromArrays(vb, ib); v.Voxelize(N)
    eye, vp = camera.default_view_proj(1280, 720)
    img = v.Render(eye, vp, 1280, 720)
    ts = []
    for _ in range(5):
        img = v.Render(eye, vp, 1280, 720)
        ts.append(v.stats()['render_ms'])
    camera.write_png(f"gpurun_out/render_{mesh}_{N}.png", img)
    print(json.dumps({"mesh": mesh, "N": N, "render_ms_1280x720": float(np.median(ts)), "opaque_px": int((img[..., 3] == 255).sum())}))
PY
exit 0
