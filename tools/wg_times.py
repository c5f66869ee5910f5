#!/usr/bin/env python3
"""Diagnostic (library built with -DDXV_QUEUE_TIMES): the time line of a launch through a PREPARED (or, plan=1,prepared=0, a kept) queue dealt
out by the hardware (k_voxelize_listed): when does every workgroup start and end, how many are in flight over time, where does the launch spend its
beginning and its end?   usage: DXV_LIBRARY=.../libdxv_qtimes.so wg_times.py [mesh] [grid] [world] [zblock] [key=value,...]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dxrvoxelizer_amd as dxv  # noqa: E402
from bench import make_mesh  # noqa: E402

mesh = sys.argv[1] if len(sys.argv) > 1 else "torus1m"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
zb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
v = dxv.Voxelizer(0)
v.set_option("lists", 2)
for kv in filter(None, (sys.argv[5] if len(sys.argv) > 5 else "").split(",")):
    v.set_option(kv.split("=")[0], int(kv.split("=")[1]))
vb, ib, _ = make_mesh(mesh)
v.InitFromArrays(vb, ib, gridDim=N)
for rank in ((0, world // 2) if world > 1 else (0,)):
    if world > 1:
        v.PrepareLaunchInterleaved(N, rank, world, zb)                  # (the launch through the prepared queue: round 6's headline path)
    for _ in range(4):
        v.VoxelizeInterleaved(N, rank, world, zb, 0) if world > 1 else v.Voxelize(N, 0)
    st = v.stats()
    n = st["plan_waves"]
    raw = np.zeros(1 << 21, np.uint64)
    v._check(v._lib.dxv_debug_download(v._ctx, 100, raw.ctypes.data_as(C.c_void_p), raw.nbytes))
    t = raw[:3 * n].reshape(n, 3)
    ran = t[:, 0] != 0
    start = t[ran, 0].astype(np.int64)
    end = (t[ran, 1] & np.uint64(0x0fffffffffffffff)).astype(np.int64)
    xcc = (t[ran, 1] >> np.uint64(60)).astype(np.int64)
    blk = np.nonzero(ran)[0]
    t0 = start.min()
    s_us, e_us = (start - t0) / 100.0, (end - t0) / 100.0
    total = e_us.max()
    edges = np.linspace(0, total, 41)
    mid = 0.5 * (edges[1:] + edges[:-1])
    inflight = [(int(((s_us <= m) & (e_us > m)).sum())) for m in mid]
    dur = e_us - s_us
    order = np.argsort(s_us)
    dec = np.array_split(order, 10)
    q = [0, 5, 25, 50, 75, 95, 99, 100]
    pct = lambda a: dict(zip(q, np.percentile(a, q).round(1).tolist()))                 # noqa: E731
    peak = max(inflight)
    print(json.dumps({"mesh": mesh, "N": N, "world": world, "zblock": zb, "rank": rank, "kernel_ms_events": round(st["voxelize_ms"], 4), "span_us": round(float(total), 1),
                      "workgroups_launched": n, "workgroups_with_a_brick": int(ran.sum()), "peak_in_flight": peak,
                      "in_flight_over_time_40_bins": inflight,
                      "us_until_90pct_of_peak": round(float(mid[next(i for i, c in enumerate(inflight) if c >= 0.9 * peak)]), 1),
                      "us_after_last_start": round(float(total - s_us.max()), 1),
                      "us_below_50pct_of_peak_at_end": round(float(total - mid[max(i for i, c in enumerate(inflight) if c >= 0.5 * peak)]), 1),
                      "brick_us_pct": pct(dur), "mean_brick_us_by_start_decile": [round(float(dur[d].mean()), 1) for d in dec],
                      "ideal_us_at_peak_concurrency": round(float(dur.sum() / peak), 1),
                      "block_mod8_equals_xcc_frac": round(float((blk % 8 == xcc).mean()), 3),
                      "end_by_xcc_us": [round(float(e_us[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)],
                      "bricks_by_xcc": [int((xcc == x).sum()) for x in range(8)],
                      "busy_us_by_xcc": [round(float(dur[xcc == x].sum() / max(peak / 8, 1)), 1) for x in range(8)]}), flush=True)
    if os.environ.get("WG_FEATURES"):
        # what makes a brick long?  per brick: its rays' texels (the cube map's, recomputed here in float32), their list lengths
        st2 = v.stats()
        R = st2["list_res"]
        cells = v.debug(dxv.voxelizer.DBG_LIST_CELLS)
        count = (cells[:, 1] & 0xffff).astype(np.int64)
        r1max = (cells[:, 1] >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
        words = t[ran, 2].astype(np.int64)
        sel = np.concatenate([np.argsort(dur)[-400:], np.random.default_rng(1).choice(len(dur), 2000, replace=False)])
        lane = np.arange(64)
        feats = []
        for i in sel:
            w_ = int(words[i]); bx, by, bz = w_ & 1023, (w_ >> 10) & 1023, w_ >> 20
            ix, iy, lz = bx * 4 + (lane & 3), by * 4 + ((lane >> 2) & 3), bz * 4 + (lane >> 4)
            iz = rank * zb + (lz // zb) * (zb * world) + (lz % zb) if world > 1 else lz
            f32 = np.float32
            o = lambda i_: ((i_.astype(f32) + f32(0.5)) / f32(N) * f32(2) - f32(1))   # noqa: E731
            ox, oy, oz = o(ix), -o(iy), o(iz)
            ax, ay, az = np.abs(ox), np.abs(oy), np.abs(oz)
            fx = (ax >= ay) & (ax >= az); fy = ~fx & (ay >= az); fz = ~fx & ~fy
            face = np.where(fx, np.where(ox < 0, 1, 0), np.where(fy, np.where(oy < 0, 3, 2), np.where(oz < 0, 5, 4)))
            u = np.where(fx, oy / ax, np.where(fy, oz / ay, ox / az)); vv = np.where(fx, oz / ax, np.where(fy, ox / ay, oy / az))
            ti = np.clip(((u + 1) * (0.5 * R)).astype(np.int64), 0, R - 1); tj = np.clip(((vv + 1) * (0.5 * R)).astype(np.int64), 0, R - 1)
            cell = (face * R + tj) * R + ti
            rho = np.sqrt(ox * ox + oy * oy + oz * oz)
            live = (count[cell] > 0) & ~(r1max[cell] < rho * 0.999)
            c = count[cell] * live
            feats.append((float(dur[i]), int(c.max()), float(c.mean()), int(len(np.unique(cell))), int(live.sum()), float(rho.min())))
        fa = np.array(feats)
        top, rnd = fa[:400], fa[400:]
        names = ["dur_us", "max_count", "mean_count", "distinct_texels", "live_rays", "rho_min"]
        print(json.dumps({"features": names, "longest_400_mean": top.mean(0).round(2).tolist(), "random_2000_mean": rnd.mean(0).round(2).tolist(),
                          "corr_with_duration_random": [round(float(np.corrcoef(rnd[:, 0], rnd[:, k])[0, 1]), 3) for k in range(1, 6)],
                          "longest_400_max_count_pct": dict(zip([5, 25, 50, 75, 95], np.percentile(top[:, 1], [5, 25, 50, 75, 95]).tolist())),
                          "random_max_count_pct": dict(zip([5, 25, 50, 75, 95, 99], np.percentile(rnd[:, 1], [5, 25, 50, 75, 95, 99]).tolist())),
                          "frac_of_random_with_max_count_over": {str(th): round(float((rnd[:, 1] > th).mean()), 4) for th in (16, 24, 32, 48, 64)},
                          "frac_of_longest_400_with_max_count_over": {str(th): round(float((top[:, 1] > th).mean()), 4) for th in (16, 24, 32, 48, 64)}}), flush=True)
