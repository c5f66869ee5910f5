#!/bin/bash
# THE evidence run of a round (one script; tools/collect_evidence.py copies the judged summaries into profiles/<round>/final/):
# suite + smoke, the bench line (default flags, the driver's flags, two ranks on one GPU over gloo, one rank through torch.distributed.run
# over RCCL), the launch disciplines side by side with the looped 8-rank shares, standalone launch times of every configuration's mesh,
# configurations and exhaustive checks, build / refit / Init timings, the reference's OBJ loader beside the product's, host-boundary
# costs, a soak, workgroup time lines (diagnostic library built HERE), rocprofv3 kernel stats of the bench command and of the builds,
# PMC passes (each in its own run, --kernel-trace only beside --pmc).  Everything lands in gpurun_out/final/.
# usage (on the GPU box): bash tools/gpu_final.sh [quick]      quick: no soak, no soup-10M PMC, fewer repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
QUICK=${1:-}
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
python bench.py --gpus 2 --backend gloo --same-device --steps 100 > $OUT/bench_2rank_same_gpu_gloo.json 2> $OUT/bench_2rank_same_gpu_gloo.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 50 --warmup 3 --interleave --no-cpu-baseline --no-extras > $OUT/bench_torchrun_world1.log 2>&1
for m in torus1m bunny16 dragon9; do python tools/prepared_ab.py $m 512 8 4 1,2,3 >> $OUT/launch_disciplines.jsonl 2>> $OUT/launch_disciplines.err; done
python tools/prepared_ab.py torus1m 256 8 4 1,2 >> $OUT/launch_disciplines.jsonl 2>> $OUT/launch_disciplines.err
python tools/prepared_ab.py dragon9 1024 8 4 1,2 >> $OUT/launch_disciplines.jsonl 2>> $OUT/launch_disciplines.err
python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m > $OUT/quick_times.jsonl 2>&1
python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 > $OUT/quick_times_256.jsonl 2>&1
python tools/quick_times.py --meshes dragon9,bunny --grid 1024 --reps 3 > $OUT/quick_times_1024.jsonl 2>&1
python tools/configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python tools/list_check_configs.py > $OUT/list_check_configs.jsonl 2>&1
python tools/build_once.py soup10m 4 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 6 > $OUT/build_torus1m.jsonl 2>&1
for m in bunny dragon dragon9 bunny16; do python tools/build_once.py $m 3 | tail -1; done > $OUT/build_other_meshes.jsonl 2>&1
python tools/build_bench.py bunny torus1m soup10m > $OUT/build_bench.jsonl 2>&1
python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
python tools/refit_loop.py bunny16 512 30 >> $OUT/refit_loop.jsonl 2>&1
python tools/init_times.py torus1m 512 3 > $OUT/init_times.jsonl 2>&1
python tools/init_times.py bunny 256 3 >> $OUT/init_times.jsonl 2>&1
python tools/obj_ingest_vs_reference.py 5 > $OUT/obj_ingest_vs_reference.jsonl 2> $OUT/obj_ingest_vs_reference.err
python tools/division_check_all.py > $OUT/division_check_all.jsonl 2>&1
python tools/ab_option.py farmap 0,1 --meshes torus1m,bunny16,dragon9,bunny --grid 512 --set lists=0 > $OUT/ab_tree_walk_brick_test.jsonl 2>&1
tools/micro/sort_check time 0 8 10 > $OUT/sort_times.jsonl 2>&1
python tools/cpu_baseline.py > $OUT/cpu_baseline.jsonl 2>&1
python tools/pcie_bench.py 512 > $OUT/pcie.jsonl 2>&1
[ -z "$QUICK" ] && python tools/gpu_soak.py 420 60606 > $OUT/soak_60606.jsonl 2>&1
# workgroup time lines: the diagnostic build of the library is made here (it does not travel: .gpurunignore)
python -c "from dxrvoxelizer_amd import build; build.build(defines=['DXV_QUEUE_TIMES'], name='qtimes')" > $OUT/build_qtimes.log 2>&1
Q=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so
if [ -f $Q ]; then
  for m in torus1m bunny16 dragon9; do DXV_LIBRARY=$Q python tools/wg_times.py $m 512 8 4 >> $OUT/wg_times.jsonl 2>> $OUT/wg_times.err; done
  DXV_LIBRARY=$Q python tools/wg_times.py torus1m 512 1 4 >> $OUT/wg_times.jsonl 2>> $OUT/wg_times.err
fi
tools/micro/l1_roof > $OUT/l1_roof.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
export DXV_WARMUP=0      # (profiles: dxv_create's warm-up launches are not the kernels these averages are about)
# the timed regions alone, so that the kernel's average over this command is the average bench.py itself reports
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
# counter passes (tools/gpu_pmc_quick.sh switches the warm-up off itself): the headline's launch (prepared), the launch that builds its
# queue, a rank's share at 2 / 4 / 8 ranks, the other 1 M-triangle mesh, the texel image, the tree walk, the soup
export PMC_LAUNCHES=5
bash tools/gpu_pmc_quick.sh torus1m torus1m 512 > $OUT/pmc_torus1m.log 2>&1
bash tools/gpu_pmc_quick.sh torus1m_unprepared torus1m 512 prepare=0 > $OUT/pmc_torus1m_unprepared.log 2>&1
bash tools/gpu_pmc_quick.sh rank8 torus1m 512 world=8 rank=0 zblock=4 > $OUT/pmc_rank8.log 2>&1
bash tools/gpu_pmc_quick.sh rank4 torus1m 512 world=4 rank=0 > $OUT/pmc_rank4.log 2>&1
bash tools/gpu_pmc_quick.sh rank2 torus1m 512 world=2 rank=0 > $OUT/pmc_rank2.log 2>&1
bash tools/gpu_pmc_quick.sh bunny16 bunny16 512 > $OUT/pmc_bunny16.log 2>&1
export PMC_LAUNCHES=3
bash tools/gpu_pmc_quick.sh treewalk torus1m 512 lists=0 > $OUT/pmc_treewalk.log 2>&1
[ -z "$QUICK" ] && bash tools/gpu_pmc_quick.sh soup10m soup10m 512 > $OUT/pmc_soup10m.log 2>&1
for t in torus1m torus1m_unprepared rank8 rank4 rank2 bunny16 treewalk soup10m; do cp gpurun_out/pmcq/$t/summary.json $OUT/pmc_${t}_summary.json 2>/dev/null; done
exit 0
