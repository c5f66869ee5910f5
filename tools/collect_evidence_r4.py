"""Copy the judged summaries of a tools/gpu_final_r4.sh run from gpurun_out/final/ (scratch) into profiles/r04/final/ (tracked)
and refresh profiles/traffic.json from the PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE, per launch of
the lists kernel over its kept work queue: one GPU, and rank 0's share of the block-cyclic partition at 2 / 4 / 8 ranks).
usage: python tools/collect_evidence_r3.py"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402  (the sources the PMC run was made on: bench.py reports the figure only while they are unchanged)
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles", "r04", "final")


def main():
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(SRC, "*.jsonl")) + glob.glob(os.path.join(SRC, "*.json")) + glob.glob(os.path.join(SRC, "*_kernel_stats.csv")) + \
            [os.path.join(SRC, n) for n in ("pytest_gpu.log", "smoke.log", "bench_torchrun_world1.log", "bench_2rank_same_gpu_gloo.err", "bench.err")]:
        if os.path.exists(f):
            shutil.copy(f, DST)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tj)) if os.path.exists(tj) else {}
    for tag, key in (("torus1m", "torus1m/512/reference/gpus1"), ("rank2", "torus1m/512/reference/gpus2"), ("rank4", "torus1m/512/reference/gpus4"),
                     ("rank8", "torus1m/512/reference/gpus8")):
        p = os.path.join(SRC, f"pmc_{tag}_summary.json")
        if not os.path.exists(p):
            continue
        s = json.load(open(p))
        k = s.get("k_voxelize_listed") or s.get("k_voxelize_queue") or s.get("k_voxelize")
        name = "k_voxelize_listed<false> (direction-space lists, one workgroup per brick of the kept work queue)" if "k_voxelize_listed" in s else \
               "k_voxelize_queue<false> (direction-space lists, persistent waves over the work queue)"
        if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
            continue
        fetch_kb, write_kb = k["FETCH_SIZE"], k["WRITE_SIZE"]
        traffic[key] = {
            "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/r04/final/pmc_{tag}_summary.json), mean per launch of the "
                      "lists kernel over its work queue (kept queue: the steps bench.py times)" + ("" if tag == "torus1m" else f" on rank 0's share of the block-cyclic partition ({tag[4:]} ranks, one GPU)") +
                      "; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B; an upper estimate for gathers); "
                      f"uncorrected total = {int((fetch_kb + write_kb) * 1024)} B",
            "kernel": name, "round": 4,
            "source_hash": source_hash()}
    with open(tj, "w") as fh:
        json.dump(traffic, fh, indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in traffic.items()}, indent=1))


if __name__ == "__main__":
    main()
