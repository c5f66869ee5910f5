#!/bin/bash
# The profile / counter part of tools/gpu_final_r5.sh on its own (into the same gpurun_out/final/).
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
tools/micro/l1_roof > $OUT/l1_roof.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
export DXV_WARMUP=0      # (profiles: dxv_create's warm-up launches are not the kernels these averages are about)
# the timed region alone, so that the kernel's average over this command is the average bench.py itself reports
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build -- python3 $GRAFT_REPO_ROOT/tools/build_once.py soup10m 3 > $OUT/prof_build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build_torus1m -- python3 $GRAFT_REPO_ROOT/tools/build_once.py torus1m 6 > $OUT/prof_build_torus1m.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_refit_loop -- python3 $GRAFT_REPO_ROOT/tools/refit_loop.py torus1m 512 20 > $OUT/prof_refit_loop.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_l1_roof -- $GRAFT_REPO_ROOT/tools/micro/l1_roof > $OUT/pmc_l1_roof.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_bench prof_build prof_build_torus1m prof_refit_loop; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
python3 tools/trace_gaps.py $OUT/prof_refit_loop 3 > $OUT/refit_loop_trace_gaps.jsonl 2>&1
python3 - > $OUT/pmc_l1_roof_summary.json <<'PY'
import csv, glob, json, os, collections
out = collections.OrderedDict()
d = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "final", "pmc_l1_roof")
rows = collections.defaultdict(dict)
for cc in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(cc)):
        if "k_gather" in r["Kernel_Name"]:
            rows[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (disp, name), c in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    clk = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if clk:
        out.setdefault(name, []).append({"line_accesses_per_clk_per_cu": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / 256 / clk, "ta_busy": c.get("TA_TA_BUSY_sum", 0) / 256 / clk, "clocks": clk})
print(json.dumps(out, indent=1))
PY
find $OUT/prof_bench $OUT/prof_build $OUT/prof_build_torus1m $OUT/prof_refit_loop $OUT/pmc_l1_roof -name "*.csv" -size +4M -delete
unset DXV_WARMUP
# counter passes (each in its own run, --kernel-trace only beside --pmc; tools/gpu_pmc_quick.sh switches the warm-up off itself)
export PMC_LAUNCHES=5
bash tools/gpu_pmc_quick.sh torus1m torus1m 512 > $OUT/pmc_torus1m.log 2>&1
bash tools/gpu_pmc_quick.sh torus1m_kept torus1m 512 plan=1 > $OUT/pmc_torus1m_kept.log 2>&1
bash tools/gpu_pmc_quick.sh rank8 torus1m 512 world=8 rank=0 zblock=4 > $OUT/pmc_rank8.log 2>&1
bash tools/gpu_pmc_quick.sh rank4 torus1m 512 world=4 rank=0 > $OUT/pmc_rank4.log 2>&1
bash tools/gpu_pmc_quick.sh rank2 torus1m 512 world=2 rank=0 > $OUT/pmc_rank2.log 2>&1
bash tools/gpu_pmc_quick.sh bunny16 bunny16 512 > $OUT/pmc_bunny16.log 2>&1
export PMC_LAUNCHES=3
bash tools/gpu_pmc_quick.sh treewalk torus1m 512 lists=0 > $OUT/pmc_treewalk.log 2>&1
bash tools/gpu_pmc_quick.sh soup10m soup10m 512 > $OUT/pmc_soup10m.log 2>&1
for t in torus1m torus1m_kept rank8 rank4 rank2 bunny16 treewalk soup10m; do cp gpurun_out/pmcq/$t/summary.json $OUT/pmc_${t}_summary.json 2>/dev/null; done
exit 0
