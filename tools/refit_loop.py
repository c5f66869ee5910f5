"""Refit-per-frame loop (the reference's PERFORM_UPDATE use, XUSGRayTracing.h:13-22): every frame uploads the vertices, refits
the tree, voxelizes -- through the tree walk (lists=0) and through lists rebuilt every frame (lists=2).  Wall-clock per frame
and the library's own stage timings.  usage: refit_loop.py [mesh] [N] [frames]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402


def main():
    mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    vb, ib, _ = make_mesh(mesh)
    vb = np.ascontiguousarray(vb, np.float32)
    v = dxv.Voxelizer(0)
    v.InitFromArrays(vb, ib)
    solid = {}
    for lists in (0, 2):
        v.set_option("lists", lists)
        for _ in range(3):
            v.UpdateVertices(vb)
            v.Voxelize(N)
        t = time.perf_counter()
        for _ in range(frames):
            v.UpdateVertices(vb)
            v.Voxelize(N)
        ms = (time.perf_counter() - t) / frames * 1e3
        st = v.stats()
        solid[lists] = v.CountSolid()
        print(json.dumps({"mesh": mesh, "N": N, "lists": lists, "frame_ms": round(ms, 3), "fps": round(1e3 / ms, 1), "refit_ms": round(st["refit_ms"], 3),
                          "list_ms": round(st["list_ms"], 3), "voxelize_ms": round(st["voxelize_ms"], 3), "entries": st["list_entries"]}))
    assert solid[0] == solid[2], solid


if __name__ == "__main__":
    main()
