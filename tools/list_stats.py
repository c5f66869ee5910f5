#!/usr/bin/env python3
"""Host replay of the lists kernel per 4x4x4 brick (tests/hostcheck: the product's own __host__ __device__ code):
what a wave of the kernel meets.  CPU only.   usage: list_stats.py MESH N R [brick-z step]"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402  (only for the bound of the mesh: test infrastructure, like this tool)
from bench import make_mesh  # noqa: E402


def main():
    name, N, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bstep = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    src, so = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp"), os.path.join(ROOT, "tests", "hostcheck", "libhostcheck.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-mavx2", "-mfma",
                           "-Wno-unknown-pragmas", "-o", so, src])
    L = C.CDLL(so)
    f32p, u32p = np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.uint32, flags="C")
    L.hc_scene_create.restype = C.c_void_p
    L.hc_scene_create.argtypes = [f32p, C.c_uint32, u32p, C.c_uint32, f32p]
    L.hc_dirmap_build.argtypes = [C.c_void_p, C.c_uint32]
    L.hc_dirmap_build.restype = C.c_uint64
    L.hc_list_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    vb, ib, _ = make_mesh(name)
    _, b = orc.bound(vb)
    h = L.hc_scene_create(np.ascontiguousarray(vb, np.float32), len(vb), np.ascontiguousarray(ib, np.uint32), len(ib) // 3, b)
    n = L.hc_dirmap_build(h, R)
    out = np.zeros(14, np.uint64)
    L.hc_list_stats(h, N, bstep, out.ctypes.data_as(C.c_void_p))
    names = ["waves", "live_lanes", "lanes_ended_by_far_radius", "waves_without_scanning_lane", "distinct_texels", "their_list_entries",
             "entries_scanned", "longest_lane_scan", "box_passes", "selected", "triangle_rounds", "hits", "distinct_selected", "hits_answered_by_class"]
    w = float(out[0])
    print(json.dumps({"mesh": name, "N": N, "R": R, "entries": int(n), "every_nth_brick_layer": bstep,
                      "per_wave": {k: round(float(out[i]) / w, 2) for i, k in enumerate(names) if i}, "waves": int(out[0])}))


if __name__ == "__main__":
    main()
