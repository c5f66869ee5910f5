#!/usr/bin/env python3
"""The ONE table of current numbers (README.md), made from the files of a round's evidence run:
    python tools/numbers_table.py [round, default r06] [--write]
--write replaces the block between `<!-- numbers:begin -->` and `<!-- numbers:end -->` in DESIGN.md and README.md.
Every figure names the file of profiles/<round>/final/ it comes from; nothing here is typed in by hand."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = next((a for a in sys.argv[1:] if not a.startswith("--")), "r06")
D = os.path.join(ROOT, "profiles", ROUND, "final")


def jl(name):
    p = os.path.join(D, name)
    out = []
    if os.path.exists(p):
        for ln in open(p):
            try:
                v = json.loads(ln)
                if isinstance(v, dict):
                    out.append(v)
            except ValueError:
                pass
    return out


def j(name):
    p = os.path.join(D, name)
    return json.load(open(p)) if os.path.exists(p) else None


def kernel_avg(csvname, substr):
    p = os.path.join(D, csvname)
    if not os.path.exists(p):
        return None
    for r in csv.DictReader(open(p)):
        if substr in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    return None


def g(x):
    return f"{x / 1e3:.1f} G"


def main():
    rows = []
    b = j("bench.json")
    if b:
        c, r = b["config"], b["roofline"]
        rows.append(("**`bench.py` `value`**: torus-1M, 512³, one `Voxelize` of a scene whose `Init` was told the grid (queue from `Init`, grid cleared and every queued brick written inside the step, one dispatch)",
                     f"**{b['ms_per_step']:.3f} ms = {g(b['value'])} voxels/s**; `roofline.frac` {r['frac']:.4f} ({r['achieved']:.0f} GB/s of algorithmic bytes over {r['kernel_ms']:.3f} ms of `{r['kernel']}`)", "bench.json"))
        d = j("bench_driver_flags.json")
        if d:
            rows.append(("… with the driver's flags (`--steps 20 --warmup 5`)", f"{d['ms_per_step']:.3f} ms = {g(d['value'])}", "bench_driver_flags.json"))
        ka = kernel_avg("prof_bench_kernel_stats.csv", "k_voxelize_listed")
        if ka:
            rows.append(("… the same command under `rocprofv3 --kernel-trace --stats`: average of `k_voxelize_listed`", f"{ka[0]:.4f} ms over {ka[1]} launches", "prof_bench_kernel_stats.csv"))
        if c.get("unprepared_step"):
            u, k = c["unprepared_step"], c["kept_step"]
            rows.append(("the same steps, queue built inside every launch (round 5's headline; `config.unprepared_step`) / queue and zeros kept (`plan = 1`, `config.kept_step`)",
                         f"{u['ms_per_step']:.3f} ms = {g(u['mvoxels_s'])} (queue build + clear {u['queue_build_ms']:.3f} ms inside) / {k['ms_per_step']:.3f} ms = {g(k['mvoxels_s'])}", "bench.json"))
        if c.get("frames_in_flight_2"):
            rows.append(("two / three voxelizations in flight on the one GPU (`config.frames_in_flight_2/3`)", f"{g(c['frames_in_flight_2']['value'])} / {g(c['frames_in_flight_3']['value'])}", "bench.json"))
        f = c.get("first_voxelize_after_init")
        if f:
            p = f["init_parts_ms"]
            rows.append(("`Init` with the grid (host wall clock) and the first three `Voxelize` calls (`config.first_voxelize_after_init`)",
                         f"{f['init_wall_ms']:.2f} ms (upload {p['upload']:.2f} + LBVH {p['lbvh']:.2f} + lists {p['lists']:.2f} + queue {p['queue']:.3f} ms of GPU time), then "
                         + " / ".join(f"{x['wall_ms']:.3f}" for x in f["voxelize_calls"]) + f" ms; the process's one warm-up pass in `dxv_create`: {c.get('warmup_ms', 0):.0f} ms", "bench.json"))
        if c.get("grid_256"):
            q = c["grid_256"]
            rows.append(("256³, the metric's other grid (`config.grid_256`): prepared / queue built inside the launch / kept / tree walk",
                         f"{q['ms']:.3f} ms = **{g(q['mvoxels_s'])}** / {g(q['unprepared_mvoxels_s'])} / {g(q['kept_queue_mvoxels_s'])} / {g(q['tree_walk_mvoxels_s'])}", "bench.json"))
        if c.get("texels"):
            t = c["texels"]
            rows.append(("N1: the reference's `R10G10B10A2` texel image written beside the grid (`config.texels`; 5 B stored per voxel)",
                         f"{t['ms']:.3f} ms = {g(t['mvoxels_s'])}; {t['achieved_gbps']:.0f} GB/s of algorithmic bytes (frac {t['frac']:.3f})", "bench.json"))
        if c.get("bunny16"):
            rows.append(("the other 1 M-triangle mesh, bunny ×16 at 512³ (standalone launches): prepared / tree walk", f"{c['bunny16']['ms']:.3f} / {c['bunny16']['tree_walk_ms']:.2f} ms", "bench.json"))
        rows.append(("tree walk (`k_voxelize`, north_star's literal kernel) / parity rule (`k_parity_rows`) on torus-1M 512³",
                     f"{c['tree_walk_ms']:.2f} ms = {g(c['tree_walk_mvoxels_s'])} / {c['other_rule']['ms']:.3f} ms = {g(c['other_rule']['mvoxels_s'])}", "bench.json"))
        if c.get("config4_dragon9_1024"):
            k4 = c["config4_dragon9_1024"]
            rows.append(("config 4's mesh and grid on one GPU: dragon ×9 at 1024³", f"{k4['ms_per_step']:.2f} ms = {g(k4['value'])}", "bench.json"))
        br = c["build_roofline"]
        rows.append(("LBVH build, 1 M triangles (`config.build_roofline`) / lists on the 512 map", f"{c['build_ms']:.3f} ms, frac {br['frac']:.3f} of the streaming roof / {c['candidates']['build_ms']:.2f} ms", "bench.json"))
        cb = b.get("cpu_baseline")
        if cb:
            rows.append(("`cpu_baseline` (the oracle's BVH tracer, kind \"port\")", f"{cb['value']:.1f} M voxels/s on {cb['cores']} threads", "bench.json"))
        src = "bench.json, profiles/traffic.json"
        if not r.get("traffic") and j("bench_after_counters.json"):      # (the counters of these sources did not exist yet when bench.json was taken: the same command afterwards)
            r = j("bench_after_counters.json")["roofline"]
            src = "bench_after_counters.json, profiles/traffic.json"
        if r.get("traffic"):
            l1 = r.get("l1") or {}
            rows.append(("`roofline.traffic` (PMC: FETCH_SIZE × 2 + WRITE_SIZE per launch) and the L1 view (`roofline.l1`)",
                         f"{r['traffic'] / 1e9:.2f} GB = {r['traffic'] / r['algorithmic_bytes_per_launch']:.1f} × the algorithmic bytes; address units {100 * l1.get('ta_busy', 0):.0f} % busy, "
                         f"{l1.get('per_brick', 0):.0f} L1 requests per brick, loads at {100 * l1.get('frac_of_load_rate', 0):.0f} % of the full-wave gather roof", src))
    s = j("pmc_torus1m_summary.json")
    if s and s.get("k_voxelize_listed"):
        k = s["k_voxelize_listed"]
        pb = k.get("per_brick", {})
        pw = k.get("per_wave", {})
        rows.append(("PMC on the headline launch, per queued brick", f"{pb.get('SQ_INSTS_VALU', 0):.0f} VALU, {pb.get('SQ_INSTS_SALU', 0):.0f} SALU, {pb.get('SQ_INSTS_VMEM_RD', 0):.1f} vector loads, "
                     f"{pb.get('SQ_INSTS_LDS', 0):.0f} LDS instructions; `SQ_WAIT_ANY` {100 * pw.get('SQ_WAIT_ANY_frac', 0):.0f} % of wave time; FETCH {k.get('FETCH_SIZE', 0) / 1024:.0f} MB, WRITE {k.get('WRITE_SIZE', 0) / 1024:.0f} MB",
                     "pmc_torus1m_summary.json"))
    for rec in jl("launch_disciplines.jsonl"):
        if rec.get("world") != 8:
            continue
        pr, un = rec.get("prepared"), rec.get("unprepared")
        if not pr:
            continue
        f1, f2 = pr["frames_1"], pr.get("frames_2")
        txt = (f"prepared: full {f1['full_ms']:.3f} ms, slowest share {f1['slowest_share_ms']:.4f} ms = **{f1['speedup_against_full_one_in_flight']:.2f} ×** with one voxelization in flight"
               + (f", {f2['speedup_against_full_one_in_flight']:.2f} × with two" if f2 else ""))
        if un:
            txt += f"; queue built inside the launch: {un['frames_1']['slowest_share_ms']:.4f} ms = {un['frames_1']['speedup_against_full_one_in_flight']:.2f} ×"
        rows.append((f"a rank's share at 8 ranks looped on ONE GPU (an estimate, not 8 GPUs): {rec['mesh']} at {rec['N']}³", txt, "launch_disciplines.jsonl"))
    qt = {(r_["mesh"], r_["N"]): r_ for n in ("quick_times.jsonl", "quick_times_256.jsonl", "quick_times_1024.jsonl") for r_ in jl(n)}
    if qt:
        cells = []
        for (m, n), r_ in sorted(qt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
            cells.append(f"{m} {n}³ {r_['prepared_ms']:.3f} / {r_['fresh_ms']:.3f} / {r_['kept_ms']:.3f} / {r_['box_ms']:.3f}" + (f" / {r_['tree_ms']:.2f}" if "tree_ms" in r_ else ""))
        rows.append(("standalone launches, median of the library's events (cold clocks): prepared / queue built inside / kept / brick box / tree walk, ms", "; ".join(cells), "quick_times*.jsonl"))
    bs = jl("build_soup10m.jsonl")
    if bs:
        soup = qt.get(("soup10m", 512))
        rows.append(("config 5, soup-10M at 512³: LBVH / lists (150 M entries, 2.4 GB) / launch", f"{bs[-1]['build_ms']:.2f} / {bs[-1]['list_ms']:.1f} ms" + (f" / {soup['prepared_ms']:.2f} ms" if soup else ""), "build_soup10m.jsonl, quick_times.jsonl"))
    rl = [r_ for r_ in jl("refit_loop.jsonl") if r_.get("mesh") == "torus1m" and r_.get("lists") == 2]
    if rl:
        fps = [r_["fps"] for r_ in rl]
        dev = [r_["fps"] for r_ in rl if r_["vertices_from"].startswith("device buffer")]
        rows.append(("N4: a mesh refitted every frame, 1 M triangles at 512³ (refit + lists rebuilt + launch per frame)",
                     f"**{min(dev):.0f} – {max(dev):.0f} frames/s** from a device buffer ({min(fps):.0f} – {max(fps):.0f} over all vertex sources)", "refit_loop.jsonl"))
    ob = jl("obj_ingest_vs_reference.jsonl")
    if ob:
        parts = []
        for f_ in sorted({r_["file"] for r_ in ob}):
            ref = next(r_ for r_ in ob if r_["file"] == f_ and r_["loader"].startswith("XUSG"))
            mine = [r_ for r_ in ob if r_["file"] == f_ and r_["loader"] == "dxv_obj_load"]
            one, best = min(mine, key=lambda r_: r_["threads"]), min(mine, key=lambda r_: r_["ms_median"])
            parts.append(f"{f_.split(' (')[0]} ({ref['file_MB']:.0f} MB): reference `ObjLoader::Import` {ref['ms_median']:.0f} ms, `dxv_obj_load` {one['ms_median']:.0f} ms on 1 thread ({one['speedup_over_reference']:.1f} ×), "
                         f"{best['ms_median']:.1f} ms on {best['threads']} ({best['speedup_over_reference']:.0f} ×), byte-identical: {all(r_['byte_identical_to_reference'] for r_ in mine)}")
        rows.append(("N2: the reference's own loader (compiled from `/root/reference`) beside the product's, same files, same box", "; ".join(parts), "obj_ingest_vs_reference.jsonl"))
    lc = jl("list_check_configs.jsonl")
    if lc:
        rows.append(("the lists' superset claim, exhaustively on the device, configs 2 – 5 + the metric's meshes", f"{sum(r_['accepted_ray_triangle_pairs'] for r_ in lc):.3g} accepted (ray, triangle) pairs, {sum(r_['violations'] for r_ in lc)} violations",
                     "list_check_configs.jsonl"))
    sk = [r_ for n in os.listdir(D) if n.startswith("soak_") for r_ in jl(n) if r_.get("soak")]
    if sk:
        rows.append(("soak against the oracle (random meshes, grids, partitions, options)", "; ".join(f"{r_['grids']} grids in {r_['seconds']:.0f} s: {r_['soak']}" for r_ in sk), "soak_*.jsonl"))
    log = os.path.join(D, "pytest_gpu.log")
    if os.path.exists(log):
        m = re.findall(r"(\d+) passed", open(log).read())
        if m:
            rows.append(("`-m gpu` suite", f"{m[-1]} passed", "pytest_gpu.log"))
    out = ["| figure (one MI355X; one run of `tools/gpu_final.sh` on one box) | value | file under `profiles/" + ROUND + "/final/` |", "|---|---|---|"]
    out += [f"| {a} | {b_} | `{c_}` |" for a, b_, c_ in rows]
    table = "\n".join(out)
    print(table)
    if "--write" in sys.argv:
        for name in ("DESIGN.md", "README.md"):
            p = os.path.join(ROOT, name)
            s = open(p).read()
            new = re.sub(r"<!-- numbers:begin -->.*?<!-- numbers:end -->", "<!-- numbers:begin -->\n" + table.replace("\\", "\\\\") + "\n<!-- numbers:end -->", s, flags=re.S)
            if new != s:
                open(p, "w").write(new)


if __name__ == "__main__":
    main()
