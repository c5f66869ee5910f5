"""Host-boundary costs around the hot path (never part of bench.py's `value`): mesh upload + build,
grid download as bytes and as device-packed bits into pageable and pinned host memory, and the
OBJ ingest at several worker counts.  One JSON line per measurement."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402


def med(f, reps=7):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts[1:]))


def main():
    import torch
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    vb, ib, _ = make_mesh("torus1m")
    v = dxv.Voxelizer(0)
    print(json.dumps({"what": "InitFromArrays (upload + LBVH build)", "tris": len(ib) // 3, "ms": med(lambda: v.InitFromArrays(vb, ib), 5)}))
    v.Voxelize(N)
    print(json.dumps({"what": "Voxelize (sync)", "N": N, "ms": med(lambda: v.Voxelize(N))}))
    nbytes = v.grid_bytes()
    pageable = np.empty(nbytes, np.uint8)
    pinned = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    lib, ctx = v._lib, v._ctx
    import ctypes as C
    for name, ptr in (("pageable", pageable.ctypes.data), ("pinned", pinned.data_ptr())):
        ms = med(lambda: v._check(lib.dxv_grid_download(ctx, C.c_void_p(ptr), nbytes)))
        print(json.dumps({"what": f"dxv_grid_download -> {name}", "MB": nbytes / 1e6, "ms": ms, "GBps": nbytes / ms / 1e6}))
    pb = (nbytes + 7) // 8
    pageable_b = np.empty(pb, np.uint8)
    pinned_b = torch.empty(pb, dtype=torch.uint8, pin_memory=True)
    for name, buf in (("pageable", pageable_b), ("pinned", pinned_b)):
        ms = med(lambda: v.GridBits(buf))
        print(json.dumps({"what": f"dxv_grid_download_packed -> {name}", "MB": pb / 1e6, "ms": ms, "voxels_GBps_equiv": nbytes / ms / 1e6}))
    assert np.array_equal(pinned_b.numpy(), np.packbits(pinned.numpy(), bitorder="little"))
    ms = med(lambda: (v.Voxelize(N, sync=False), v.GridBits(pinned_b)))
    print(json.dumps({"what": "Voxelize + packed download to pinned host memory", "N": N, "ms": ms, "Mvox_per_s": N ** 3 / ms / 1e3}))
    ms = med(lambda: (v.Voxelize(N, sync=False), v._check(lib.dxv_grid_download(ctx, C.c_void_p(pinned.data_ptr()), nbytes))))
    print(json.dumps({"what": "Voxelize + byte download to pinned host memory", "N": N, "ms": ms, "Mvox_per_s": N ** 3 / ms / 1e3}))

    # OBJ ingest: the torus as an OBJ text file
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from test_obj_ingest import load_with_threads, write_torus_obj
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "torus1m.obj")
        write_torus_obj(path, 1000, 500, False)
        size = os.path.getsize(path)
        for threads in (1, 2, 4, 8, 16):
            ms = med(lambda: load_with_threads(path, threads), 5)
            print(json.dumps({"what": "dxv_obj_load", "threads": threads, "file_MB": size / 1e6, "tris": 1000000, "ms": ms, "MBps": size / ms / 1e3}))
        ms = med(lambda: dxv.obj_load(path), 5)
        print(json.dumps({"what": "dxv_obj_load", "threads": "default", "ms": ms, "MBps": size / ms / 1e3}))


if __name__ == "__main__":
    main()
