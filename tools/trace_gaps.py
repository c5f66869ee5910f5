#!/usr/bin/env python3
"""Where a frame of the refit loop goes: from a rocprofv3 --kernel-trace CSV, the kernels of the last frames in order with
the idle time in front of each.  usage: trace_gaps.py <dir with *_kernel_trace.csv> [frames=5] [anchor kernel substring]"""
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_voxelize_queue"
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
sel = idx[-frames - 1:]
out = []
for a, b in zip(sel[:-1], sel[1:]):
    seq = rows[a + 1:b + 1]
    t0 = rows[a][1]
    fr = {"frame_us": round((rows[b][1] - rows[a][1]) / 1e3, 1), "busy_us": round(sum(e - s for s, e, _ in seq) / 1e3, 1), "kernels": []}
    prev = t0
    for s, e, n in seq:
        fr["kernels"].append([n[:40], round((s - prev) / 1e3, 1), round((e - s) / 1e3, 1)])          # name, gap in front, duration
        prev = e
    out.append(fr)
last = out[-1]
print(json.dumps({"file": os.path.basename(f), "frames": [{k: v for k, v in fr.items() if k != "kernels"} for fr in out]}))
print(json.dumps({"last_frame": last["kernels"]}))
big = [[n, g] for n, g, _ in last["kernels"] if g > 3.0]
print(json.dumps({"gaps_over_3us_in_front_of": big, "sum_gaps_us": round(sum(g for _, g, _ in last["kernels"]), 1)}))
