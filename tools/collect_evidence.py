"""Copy the judged summaries of a tools/gpu_final.sh run from gpurun_out/final/ (scratch) into
profiles/r01/final/ (tracked) and refresh profiles/traffic.json.  For every rocprofv3 output
directory the newest run is taken.  usage: python tools/collect_evidence.py [round_dir]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "gpurun_out", "final")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
DST = os.path.join(ROOT, "profiles", ROUND, "final")


def newest(pattern):
    files = glob.glob(pattern)            # (run directories are named after a PID: the newest is the last one written)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    for k in ("k_voxelize_redo", "k_voxelize", "k_parity_rows", "k_count"):
        if "dxv::" + k + "<" in name or "dxv::" + k + "(" in name:
            return k
    return None


def main():
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(SRC, "*.jsonl")) + glob.glob(os.path.join(SRC, "*.json")) + \
            [os.path.join(SRC, n) for n in ("pytest_gpu.log", "smoke.log", "bench_torchrun_world1.log", "bench_2rank_same_gpu_gloo.err")]:
        if os.path.exists(f):
            shutil.copy(f, DST)
    for png in glob.glob(os.path.join(SRC, "render_*.png")):
        shutil.copy(png, os.path.join(DST, ".."))
    ks = newest(os.path.join(SRC, "prof_bench", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(DST, "bench_kernel_stats.csv"))
    ks = newest(os.path.join(SRC, "prof_refit_loop", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks, os.path.join(DST, "refit_loop_kernel_stats.csv"))
    summary = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        cc = newest(os.path.join(d, "*", "*_counter_collection.csv"))
        if not cc:
            continue
        shutil.copy(cc, os.path.join(DST, os.path.basename(d) + ".csv"))
        per_dispatch = defaultdict(dict)
        with open(cc) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k:
                    per_dispatch[(k, row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
        parity = os.path.basename(d).endswith("_parity")
        for (k, _), counters in per_dispatch.items():
            if parity and k != "k_parity_rows":
                continue
            if not parity and k == "k_parity_rows":
                continue
            for name, v in counters.items():
                summary[k][name].append(v)
    out = {"workload": "torus1m/512, per launch (mean over the profiled launches)", "kernels": {}}
    for k, counters in summary.items():
        out["kernels"][k] = {name: sum(v) / len(v) for name, v in counters.items()}
    for k, m in out["kernels"].items():
        w = m.get("SQ_WAVES")
        if w:
            m["per_wave"] = {n: round(m[n] / w, 1) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS",
                                                               "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_WAVE_CYCLES") if n in m}
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in m and m.get("SQ_WAVE_CYCLES"):
                    m["per_wave"][n + "_frac"] = round(m[n] / m["SQ_WAVE_CYCLES"], 3)
    with open(os.path.join(DST, "pmc_summary.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    kv = out["kernels"].get("k_voxelize", {})
    if "FETCH_SIZE" in kv and "WRITE_SIZE" in kv:
        kc = out["kernels"].get("k_count", {})
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = json.load(open(tj)) if os.path.exists(tj) else {}
        fetch_kb, write_kb = kv["FETCH_SIZE"], kv["WRITE_SIZE"]
        traffic["torus1m/512/reference/gpus1"] = {
            "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
            "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{ROUND}/final/pmc_fetch.csv, "
                      "pmc_write.csv), mean per k_voxelize launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                      "128-B requests as 64 B): calibrated in the same pass on k_count, a 16-B/lane streaming read of exactly "
                      f"134,217,728 B, which reports {kc.get('FETCH_SIZE', float('nan')):.0f} KB = 1/2. The gathers of k_voxelize are "
                      "not a streaming pattern, so the doubled figure is an upper estimate; uncorrected total = "
                      f"{int((fetch_kb + write_kb) * 1024)} B",
            "kernel": "k_voxelize<Brick<4,4,4>,16,0,false,4,0> (direction-space lists; the tree walk's figures: profiles/r01/final/tree_walk/)",
            "round": int(ROUND[1:])}
        with open(tj, "w") as fh:
            json.dump(traffic, fh, indent=1)
    print(json.dumps({k: {n: (round(v, 1) if isinstance(v, float) else v) for n, v in c.items()} for k, c in out["kernels"].items()}, indent=1)[:3000])


if __name__ == "__main__":
    main()
