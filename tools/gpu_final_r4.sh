#!/bin/bash
# Round 4 evidence run: suite, smoke, bench line (+ rocprof kernel stats of the same command), multi-rank plumbing on one GPU,
# PMC passes (queue kernel: kept queue and rebuilt every launch, a rank's share at 2 / 4 / 8 ranks, bunny x16, tree walk, soup-10M),
# rank times, configurations, exhaustive list and queue checks, build timings, refit loops, soak.  Everything lands in gpurun_out/final/.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
python bench.py --gpus 2 --backend gloo --same-device --steps 100 > $OUT/bench_2rank_same_gpu_gloo.json 2> $OUT/bench_2rank_same_gpu_gloo.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 50 --warmup 3 --interleave --no-cpu-baseline --no-extras > $OUT/bench_torchrun_world1.log 2>&1
for m in torus1m bunny16; do python tools/rank_times.py $m 512 lists=2 noparity zb8 >> $OUT/rank_times.jsonl 2>&1; done
python tools/rank_times.py dragon9 1024 lists=2 noparity zb8 >> $OUT/rank_times.jsonl 2>&1
python tools/rank_times.py torus1m 512 lists=2,plan=2 noparity zb8 > $OUT/rank_times_rebuilt_every_launch.jsonl 2>&1
python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m --tree --fresh > $OUT/quick_times.jsonl 2>&1
python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --tree --fresh > $OUT/quick_times_256.jsonl 2>&1
python tools/quick_times.py --meshes dragon9,bunny --grid 1024 --reps 3 --fresh > $OUT/quick_times_1024.jsonl 2>&1
python tools/quick_times.py --meshes soup10m,torus1m,dragon9 --frames 3 --reps 5 > $OUT/frames3.jsonl 2>&1
python tools/configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
python tools/list_check_configs.py > $OUT/list_check_configs.jsonl 2>&1
python tools/build_once.py soup10m 4 > $OUT/build_soup10m.jsonl 2>&1
python tools/build_once.py torus1m 4 > $OUT/build_torus1m.jsonl 2>&1
python tools/build_bench.py bunny torus1m soup10m > $OUT/build_bench.jsonl 2>&1
python tools/refit_loop.py torus1m 512 60 > $OUT/refit_loop.jsonl 2>&1
python tools/refit_loop.py bunny16 512 30 >> $OUT/refit_loop.jsonl 2>&1
python tools/frame_loop.py > $OUT/frame_loop.jsonl 2>&1
python tools/cpu_baseline.py > $OUT/cpu_baseline.jsonl 2>&1
python tools/pcie_bench.py 512 > $OUT/pcie.jsonl 2>&1
python tools/gpu_soak.py 300 40404 > $OUT/soak_40404.jsonl 2>&1
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so python tools/queue_times.py torus1m 512 > $OUT/queue_wave_times.jsonl 2>&1
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so QT_WORLD=8 python tools/queue_times.py torus1m 512 >> $OUT/queue_wave_times.jsonl 2>&1
cd /tmp && export TMPDIR=/tmp
# the timed region alone, so that the kernel's average over this command is the average bench.py itself reports
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build -- python3 $GRAFT_REPO_ROOT/tools/build_once.py soup10m 3 > $OUT/prof_build.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_refit_loop -- python3 $GRAFT_REPO_ROOT/tools/refit_loop.py torus1m 512 20 > $OUT/prof_refit_loop.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_bench prof_build prof_refit_loop; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
# where a frame of the refit loop goes (kernels of the last frames in order, idle time in front of each): the loop's last variant, nothing waited for
python3 tools/trace_gaps.py $OUT/prof_refit_loop 3 > $OUT/refit_loop_trace_gaps.jsonl 2>&1
find $OUT/prof_bench $OUT/prof_build $OUT/prof_refit_loop -name "*.csv" -size +4M -delete
# counter passes (each in its own run, --kernel-trace only beside --pmc)
export PMC_LAUNCHES=5
bash tools/gpu_pmc_quick.sh torus1m torus1m 512 > $OUT/pmc_torus1m.log 2>&1
bash tools/gpu_pmc_quick.sh torus1m_fresh torus1m 512 plan=2 > $OUT/pmc_torus1m_fresh.log 2>&1
bash tools/gpu_pmc_quick.sh rank8 torus1m 512 world=8 rank=0 zblock=4 > $OUT/pmc_rank8.log 2>&1
bash tools/gpu_pmc_quick.sh rank4 torus1m 512 world=4 rank=0 > $OUT/pmc_rank4.log 2>&1
bash tools/gpu_pmc_quick.sh rank2 torus1m 512 world=2 rank=0 > $OUT/pmc_rank2.log 2>&1
bash tools/gpu_pmc_quick.sh bunny16 bunny16 512 > $OUT/pmc_bunny16.log 2>&1
export PMC_LAUNCHES=3
bash tools/gpu_pmc_quick.sh treewalk torus1m 512 lists=0 > $OUT/pmc_treewalk.log 2>&1
bash tools/gpu_pmc_quick.sh soup10m soup10m 512 > $OUT/pmc_soup10m.log 2>&1
for t in torus1m torus1m_fresh rank8 rank4 rank2 bunny16 treewalk soup10m; do cp gpurun_out/pmcq/$t/summary.json $OUT/pmc_${t}_summary.json 2>/dev/null; done
exit 0
