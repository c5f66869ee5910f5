#!/usr/bin/env python3
"""The one same-box baseline of REFERENCE code this project can have: XUSG::ObjLoader::Import (XUSG/Optional/XUSGObjLoader.cpp:18-40,
compiled from /root/reference into oracle/_ref/ref_objloader by oracle/Makefile) timed beside the product's dxv_obj_load on the
same files, on the same host cores -- and the two outputs compared byte for byte at that size too.
Files: torus-1M (1000 x 500 quads, no vn: the loader computes the normals) and bunny x16 (1,114,656 triangles) written as OBJ text.
    python tools/obj_ingest_vs_reference.py [runs]        one JSON line per (file, loader, threads); needs no GPU"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dxrvoxelizer_amd as dxv  # noqa: E402
from dxrvoxelizer_amd import meshes  # noqa: E402
from oracle import orc  # noqa: E402  (the checker's binding of oracle/_ref/ref_objloader: this tool is test infrastructure)
from test_obj_ingest import load_with_threads, write_torus_obj  # noqa: E402

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_objloader")


def write_mesh_obj(path, vb, ib):
    """positions and triangles of an indexed mesh as OBJ text (no vn: both loaders compute the normals)"""
    with open(path, "w") as f:
        f.write("# %d vertices, %d triangles\n" % (len(vb), len(ib) // 3))
        np.savetxt(f, vb[:, :3], fmt="v %.9g %.9g %.9g")
        np.savetxt(f, ib.reshape(-1, 3).astype(np.int64) + 1, fmt="f %d %d %d")


def time_c_call(path, threads):
    """wall ms of dxv_obj_load itself (no copy into numpy arrays)"""
    import ctypes as C
    lib = dxv.load_library()
    vb, ib = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, ni = C.c_uint32(), C.c_uint32()
    aabb = np.zeros(6, np.float32)
    old = os.environ.get("DXV_OBJ_THREADS")
    os.environ["DXV_OBJ_THREADS"] = str(threads)
    try:
        t0 = time.perf_counter()
        rc = lib.dxv_obj_load(os.fsencode(path), C.byref(vb), C.byref(nv), C.byref(ib), C.byref(ni), aabb)
        ms = (time.perf_counter() - t0) * 1e3
    finally:
        if old is None:
            del os.environ["DXV_OBJ_THREADS"]
        else:
            os.environ["DXV_OBJ_THREADS"] = old
    assert rc == 0
    lib.dxv_free(vb)
    lib.dxv_free(ib)
    return ms


def main():
    if not os.path.exists(EXE):
        raise SystemExit("oracle/_ref/ref_objloader is missing: `make -C oracle ref` where /root/reference exists")
    cores = orc.usable_cores()
    with tempfile.TemporaryDirectory() as d:
        files = []
        p = os.path.join(d, "torus1m.obj")
        write_torus_obj(p, 1000, 500, False)
        files.append(("torus-1M (1000 x 500 quads)", p))
        b = np.load(os.path.join(ROOT, "tests", "golden", "meshes", "bunny.npz"))
        vb, ib = meshes.midpoint_subdivide(b["vb"], b["ib"], 2)
        p = os.path.join(d, "bunny16.obj")
        write_mesh_obj(p, vb, ib)
        files.append(("bunny x16", p))
        for label, path in files:
            size = os.path.getsize(path)
            # the reference's loader: Import alone, timed inside the process
            out = subprocess.run([EXE, path, os.path.join(d, "ref.bin"), str(RUNS)], check=True, capture_output=True, text=True).stdout
            ref_ms = [json.loads(ln)["ms"] for ln in out.splitlines() if ln.startswith("{")]
            rvb, rib, raabb = orc.ref_objloader(path)
            rec = {"file": label, "file_MB": round(size / 1e6, 1), "tris": len(rib) // 3, "verts": len(rvb), "host_cores": cores}
            print(json.dumps({**rec, "loader": "XUSG::ObjLoader::Import (reference, compiled here)", "threads": 1, "runs": RUNS,
                              "ms_median": float(np.median(ref_ms)), "ms_min": float(np.min(ref_ms)), "MBps": size / float(np.median(ref_ms)) / 1e3}), flush=True)
            for threads in sorted({1, min(8, cores), cores}):
                ts = []
                for _ in range(RUNS):
                    ts.append(time_c_call(path, threads))              # (the C call alone, like Import alone above)
                pvb, pib, paabb = load_with_threads(path, threads)
                same = np.array_equal(pvb.view(np.uint32), rvb.view(np.uint32)) and np.array_equal(pib, rib) and np.array_equal(paabb.view(np.uint32), raabb.view(np.uint32))
                print(json.dumps({**rec, "loader": "dxv_obj_load", "threads": threads, "runs": RUNS, "ms_median": float(np.median(ts)), "ms_min": float(np.min(ts)),
                                  "MBps": size / float(np.median(ts)) / 1e3, "speedup_over_reference": float(np.median(ref_ms)) / float(np.median(ts)),
                                  "byte_identical_to_reference": bool(same)}), flush=True)
                assert same, "the product's loader output differs from the reference's"


if __name__ == "__main__":
    main()
