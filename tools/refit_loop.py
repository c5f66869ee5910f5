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
    v.InitDynamic(vb, ib)                    # (a mesh that is refitted every frame: LBVH only, the lists are built per frame)
    # the vertices once on the GPU as well (a mesh animated there never crosses PCIe): a plain hipMalloc through ctypes, no torch
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    dvb = C.c_void_p()
    assert hip.hipMalloc(C.byref(dvb), vb.nbytes) == 0 and hip.hipMemcpy(dvb, vb.ctypes.data_as(C.c_void_p), vb.nbytes, 1) == 0
    solid = {}
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    for lists, device in ((0, False), (2, False), (2, "overlap"), (2, "overlap, pinned"), (2, True), (2, "device, queued")):
        v.set_option("lists", lists)

        def update():
            v.UpdateVerticesDevice(dvb.value, len(vb)) if device in (True, "device, queued") else v.UpdateVertices(vb)

        for _ in range(3):
            update()
            v.Voxelize(N)
        t = time.perf_counter()
        pinned = device == "overlap, pinned"
        if pinned:                                                       # page-locked: the copy is one DMA, no staging through the runtime
            assert hip.hipHostRegister(vb.ctypes.data_as(C.c_void_p), vb.nbytes, 0) == 0
        if device in ("overlap", "overlap, pinned"):
            # the next frame's vertices cross PCIe while this frame's launch runs (dxv_update_vertices does not wait for the
            # frames; dxv_refit does)
            for _ in range(frames):
                v.Voxelize(N, sync=False)
                v.UpdateVertices(vb, refit=False)
                v.Refit()
            v.Voxelize(N)
        elif device == "device, queued":
            # the application's loop when the grid is consumed on the GPU: nothing waits for a launch -- the copy, the refit and the
            # lists' counting pass queue up behind it, and the host's one wait per frame (root box and entry total) is dxv_refit's
            for _ in range(frames):
                update()
                v.Voxelize(N, sync=False)
            v.Sync()
        else:
            for _ in range(frames):
                update()
                v.Voxelize(N)
        ms = (time.perf_counter() - t) / frames * 1e3
        if pinned:
            v.SyncAll()
            hip.hipHostUnregister(vb.ctypes.data_as(C.c_void_p))
        st = v.stats()
        solid[(lists, device)] = v.CountSolid()
        where = {False: "host array (12 MB over PCIe per frame at 1 M triangles)", True: "device buffer", "device, queued": "device buffer, launches not waited for",
                 "overlap": "host array, uploaded while the previous frame's launch runs",
                 "overlap, pinned": "page-locked host array, uploaded while the previous frame's launch runs"}[device]
        print(json.dumps({"mesh": mesh, "N": N, "lists": lists, "vertices_from": where,
                          "frame_ms": round(ms, 3), "fps": round(1e3 / ms, 1), "refit_ms": round(st["refit_ms"], 3),
                          "list_ms": round(st["list_ms"], 3), "voxelize_ms": round(st["voxelize_ms"], 3), "entries": st["list_entries"]}))
    assert len(set(solid.values())) == 1, solid


if __name__ == "__main__":
    main()
