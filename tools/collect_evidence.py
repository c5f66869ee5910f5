"""Copy the judged summaries of a tools/gpu_final.sh run from gpurun_out/final/ (scratch) into profiles/<round>/final/ (tracked)
and refresh profiles/traffic.json from the PMC passes: HBM bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md +
WRITE_SIZE) and the vector-L1 view of the same launch (line accesses, L2 read requests, address-unit busy cycles per clock and
CU, against the measured roof of tools/micro/l1_roof.hip) of the kernel bench.py's headline runs -- the lists kernel dealt out by
the hardware over the queue Init prepared, the grid's clear riding in the same dispatch (one GPU, and rank 0's share of the
block-cyclic partition at 2 / 4 / 8 ranks).
usage: python tools/collect_evidence.py [round, default r06]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402  (the sources the PMC run was made on: bench.py reports the figures only while they are unchanged)
SRC = os.path.join(ROOT, "gpurun_out", "final")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
DST = os.path.join(ROOT, "profiles", ROUND, "final")
CUS = 256


def roof():
    """lines per clock and CU the L1 / address units sustain for a gather of 16 B per lane (tools/micro/l1_roof.hip on the same box:
    l1-resident and L2-resident tables), at the clock the PMC run itself saw"""
    p = os.path.join(SRC, "l1_roof.jsonl")
    out = {}
    if os.path.exists(p):
        for ln in open(p):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if d.get("pattern", "").startswith("lines"):
                out[d["pattern"]] = {"line_accesses_per_ns_per_cu": d["line_accesses_per_ns_per_cu"], "loads_per_ns_per_cu": d["loads_per_ns_per_cu"]}
    return out


def main():
    os.makedirs(DST, exist_ok=True)
    for f in glob.glob(os.path.join(SRC, "*.jsonl")) + glob.glob(os.path.join(SRC, "*.json")) + glob.glob(os.path.join(SRC, "*_kernel_stats.csv")) + \
            [os.path.join(SRC, n) for n in ("pytest_gpu.log", "smoke.log", "bench_torchrun_world1.log", "bench_2rank_same_gpu_gloo.err", "bench.err", "obj_ingest_vs_reference.err")]:
        if os.path.exists(f):
            shutil.copy(f, DST)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = {}
    micro = roof()
    for tag, key in (("torus1m", "torus1m/512/reference/gpus1"), ("rank2", "torus1m/512/reference/gpus2"), ("rank4", "torus1m/512/reference/gpus4"),
                     ("rank8", "torus1m/512/reference/gpus8")):
        p = os.path.join(SRC, f"pmc_{tag}_summary.json")
        if not os.path.exists(p):
            continue
        s = json.load(open(p))
        k = s.get("k_voxelize_listed") or s.get("k_voxelize_queue")
        which = "k_voxelize_listed" if s.get("k_voxelize_listed") else "k_voxelize_queue"
        if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
            continue
        fetch_kb, write_kb = k["FETCH_SIZE"], k["WRITE_SIZE"]
        ent = {
            "hbm_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024), "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
            "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{ROUND}/final/pmc_{tag}_summary.json), mean per launch of the "
                      "lists kernel dealt out by the hardware over the prepared queue, the clear of the unqueued bricks in the same dispatch (the steps bench.py times)" +
                      ("" if tag == "torus1m" else f" on rank 0's share of the block-cyclic partition ({tag[4:]} ranks, one GPU)") +
                      "; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B; an upper estimate for gathers); "
                      f"uncorrected total = {int((fetch_kb + write_kb) * 1024)} B",
            "kernel": which + "<false> (direction-space lists; prepared queue from Init, one workgroup per queued brick + the clearing workgroups)", "round": int(ROUND[1:]) if ROUND[1:].isdigit() else ROUND,
            "source_hash": source_hash()}
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in k and "GRBM_GUI_ACTIVE" in k:
            clk = k["GRBM_GUI_ACTIVE"] / 8.0                              # (the counter sums the eight XCDs)
            ms = k.get("profiled_ms_mean")
            ghz = clk / (ms * 1e6) if ms else None
            acc = k["TCP_TOTAL_CACHE_ACCESSES_sum"]
            l1 = {"line_accesses": acc, "l2_read_requests": k.get("TCP_TCC_READ_REQ_sum"), "vector_loads": k.get("SQ_INSTS_VMEM_RD"),
                  "lines_per_load": acc / k["SQ_INSTS_VMEM_RD"] if k.get("SQ_INSTS_VMEM_RD") else None,
                  "per_brick": (k.get("per_brick") or {}).get("TCP_TOTAL_CACHE_ACCESSES_sum"),
                  "clocks_per_xcd": clk, "clock_ghz_in_the_pmc_run": ghz,
                  "per_clk_per_cu": acc / CUS / clk, "l2_read_requests_per_clk_per_cu": (k.get("TCP_TCC_READ_REQ_sum") or 0) / CUS / clk,
                  "loads_per_clk_per_cu": (k.get("SQ_INSTS_VMEM_RD") or 0) / CUS / clk,
                  "ta_busy": k.get("TA_TA_BUSY_sum", 0) / CUS / clk}
            if micro and ghz:
                m64, m32, l2 = micro.get("lines64_l1"), micro.get("lines32_l1"), micro.get("lines32_l2")
                if m64:                                                  # the review's roof: a gather of 16 B per lane to 64 different L1-resident lines
                    l1["peak"] = m64["line_accesses_per_ns_per_cu"] / ghz
                    l1["frac"] = l1["per_clk_per_cu"] / l1["peak"]
                # what the micro-benchmark shows beside it: the pipe is bound by LOADS, not lines -- one 16-byte-per-lane load per ~48 clocks
                # and CU whether its lanes name 32 lines or 64 -- so the kernel's own roof is its lines-per-load x that load rate; its
                # misses are bounded by the L2 -> L1 fill rate
                if m32:
                    peak_loads = m32["loads_per_ns_per_cu"] / ghz
                    l1["peak_loads_per_clk_per_cu"] = peak_loads
                    l1["frac_of_load_rate"] = l1["loads_per_clk_per_cu"] / peak_loads
                if l2:
                    l1["peak_l2_fills_per_clk_per_cu"] = l2["line_accesses_per_ns_per_cu"] / ghz
                    l1["frac_of_l2_fill_rate"] = l1["l2_read_requests_per_clk_per_cu"] / l1["peak_l2_fills_per_clk_per_cu"]
                l1["roof_source"] = "tools/micro/l1_roof.hip on the box of the PMC run (profiles/" + ROUND + "/final/l1_roof.jsonl): 7 waves per SIMD, 16 B per lane"
            ent["l1"] = l1
        traffic[key] = ent
    with open(tj, "w") as fh:
        json.dump(traffic, fh, indent=1)
    print(json.dumps({k: {"hbm": v["hbm_bytes_per_launch"], "l1": v.get("l1")} for k, v in traffic.items()}, indent=1))


if __name__ == "__main__":
    main()
