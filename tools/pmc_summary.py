#!/usr/bin/env python3
"""Per-launch and per-wave means of the PMC passes of tools/gpu_pmc_quick.sh for the voxelize kernels."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kind(name):
    """the lists kernel through the work queue -- persistent waves (a queue's first launch, plan = 2) or one workgroup per queued brick
    (a kept queue of known size) -- and the kernels launched over a brick box are different kernels"""
    return "k_voxelize_queue" if "k_voxelize_queue" in name else "k_voxelize_listed" if "k_voxelize_listed" in name else "k_voxelize"


def main():
    d = sys.argv[1]
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for cc in glob.glob(os.path.join(d, "*", "*", "*_counter_collection.csv")):
        disp = defaultdict(dict)
        with open(cc) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "dxv::k_voxelize" not in name or "redo" in name:
                    continue
                disp[(row["Dispatch_Id"], kind(name))][row["Counter_Name"]] = float(row["Counter_Value"])
        for (_, kd), c in disp.items():
            for k, v in c.items():
                per[kd][k].append(v)
    for kt in glob.glob(os.path.join(d, "*", "*", "*_kernel_trace.csv")):
        with open(kt) as fh:
            for row in csv.DictReader(fh):
                if "dxv::k_voxelize" in row["Kernel_Name"] and "redo" not in row["Kernel_Name"]:
                    dur[kind(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    out = {}
    for k, c in per.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        w = m.get("SQ_WAVES", 0) or 1
        m["per_wave"] = {n: round(m[n] / w, 1) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS",
                                                           "TCP_TOTAL_CACHE_ACCESSES_sum") if n in m}
        if "SQ_WAVE_CYCLES" in m:
            m["per_wave"]["wave_quad_cycles"] = round(m["SQ_WAVE_CYCLES"] / w, 1)
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in m:
                    m["per_wave"][n + "_frac"] = round(m[n] / m["SQ_WAVE_CYCLES"], 3)
        # persistent waves: per BRICK is the figure that compares with a wave of the brick-box launch (PMC_BRICKS: queued bricks of the launch)
        bricks = float(os.environ.get("PMC_BRICKS", "0") or 0)
        if k in ("k_voxelize_queue", "k_voxelize_listed") and bricks:
            m["bricks"] = bricks
            m["per_brick"] = {n: round(m[n] / bricks, 1) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
                                                                    "TCP_TOTAL_CACHE_ACCESSES_sum") if n in m}
        if dur.get(k):
            m["profiled_ms_mean"] = sum(dur[k]) / len(dur[k])
        out[k] = m
    with open(os.path.join(d, "summary.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: {"per_wave": v.get("per_wave"), "per_brick": v.get("per_brick"), "ms": v.get("profiled_ms_mean"), "TA_BUSY": v.get("TA_TA_BUSY_sum"),
                          "GRBM": v.get("GRBM_GUI_ACTIVE"), "waves": v.get("SQ_WAVES")} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
