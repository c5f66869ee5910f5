"""The reference's own loop (DXRVoxelizer.cpp: OnUpdate + OnRender, its window-title FPS covers
voxelisation AND the 1280x720 volume ray-march): frames per second of Voxelize + Render here, for the
reference's default scene (bunny, 64^3) and for the metric's scene, static and with animated vertices
(UpdateVertices = upload + refit every frame)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from dxrvoxelizer_amd import camera  # noqa: E402
from bench import make_mesh  # noqa: E402


def loop(v, N, vb, animate, frames, dvb=None):
    eye, vp = camera.default_view_proj(1280, 720)
    for _ in range(3):
        v.Voxelize(N)
        v.Render(eye, vp, 1280, 720)
    t = time.perf_counter()
    for f in range(frames):
        if animate and dvb is not None:
            v.UpdateVerticesDevice(dvb.data_ptr(), len(vb))      # vertices animated on the GPU: device-to-device copy + refit
        elif animate:
            v.UpdateVertices(vb)                    # same positions: the cost of upload + refit is what counts
        v.Voxelize(N)
        v.Render(eye, vp, 1280, 720)
    return (time.perf_counter() - t) / frames * 1e3


def main():
    import torch
    torch.cuda.init()                               # (before the library's own HIP runtime comes up: torch ships its own copy)
    v = dxv.Voxelizer(0)
    for mesh, N, frames in (("bunny", 64, 300), ("bunny", 256, 200), ("torus1m", 512, 60)):
        vb, ib, _ = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        dvb = torch.from_numpy(np.ascontiguousarray(vb, np.float32)).cuda()
        torch.cuda.synchronize()
        for animate, skip in ((False, 1), (True, 1), ("device", 1), (False, 0)):
            v.set_option("skipempty", skip)
            ms = loop(v, N, np.ascontiguousarray(vb, np.float32), bool(animate), frames, dvb if animate == "device" else None)
            st = v.stats()
            print(json.dumps({"scene": mesh, "N": N, "animated_vertices": animate, "skipempty": skip, "frame_ms": round(ms, 3), "fps": round(1e3 / ms, 1),
                              "voxelize_ms": round(st["voxelize_ms"], 3), "render_ms": round(st["render_ms"], 3),
                              "refit_ms": round(st["refit_ms"], 3) if animate else None, "list_ms": round(st["list_ms"], 3) if animate else None}))


if __name__ == "__main__":
    main()
