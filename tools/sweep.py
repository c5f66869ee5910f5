#!/usr/bin/env python3
"""Kernel-variant sweep on one GPU: voxelize time (HIP events, median of R) for every brick shape /
stack depth on the named meshes and grids.  Prints one JSON line per configuration.

    python tools/sweep.py [--meshes torus1m,bunny] [--grids 256,512] [--bricks 0,1,2,3] [--stacks 0,64] [--reps 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--meshes", default="torus1m")
    ap.add_argument("--grids", default="256,512")
    ap.add_argument("--bricks", default="0,1,2,3")
    ap.add_argument("--stacks", default="0")
    ap.add_argument("--modes", default="reference")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opts", default="", help="extra options key=value,key=value applied to every run")
    args = ap.parse_args()

    import numpy as np

    import dxrvoxelizer_amd as dxv
    from bench import make_mesh

    v = dxv.Voxelizer(0)
    for kv in filter(None, args.opts.split(",")):
        k, val = kv.split("=")
        v.set_option(k, int(val))
    for mesh in args.meshes.split(","):
        vb, ib, label = make_mesh(mesh)
        v.InitFromArrays(vb, ib)
        st = v.stats()
        print(json.dumps({"mesh": mesh, "tris": st["num_tris"], "tree_height": st["tree_height"],
                          "build_ms": st["build_ms"], "prep_ms": st["prep_ms"], "sort_ms": st["sort_ms"],
                          "hierarchy_ms": st["hierarchy_ms"], "refit_ms": st["refit_ms"],
                          "upload_ms": st["upload_ms"]}), flush=True)
        for N in (int(x) for x in args.grids.split(",")):
            for mode in args.modes.split(","):
                m = dxv.MODE_REFERENCE if mode == "reference" else dxv.MODE_PARITY
                for brick in (int(x) for x in args.bricks.split(",")):
                    for stack in (int(x) for x in args.stacks.split(",")):
                        v.set_option("brick", brick)
                        v.set_option("stack", stack)
                        ts = []
                        try:
                            v.Voxelize(N, m)            # warm-up
                            for _ in range(args.reps):
                                v.Voxelize(N, m)
                                ts.append(v.stats()["voxelize_ms"])
                            ms = float(np.median(ts))
                            print(json.dumps({"mesh": mesh, "N": N, "mode": mode, "brick": brick,
                                              "stack": v.stats()["stack_entries"], "ms": ms, "min_ms": min(ts),
                                              "mvox_s": N ** 3 / ms / 1e3, "solid": v.CountSolid()}), flush=True)
                        except dxv.DxvError as e:
                            print(json.dumps({"mesh": mesh, "N": N, "mode": mode, "brick": brick, "stack": stack,
                                              "error": str(e)}), flush=True)
    v.close()


if __name__ == "__main__":
    main()
