#!/bin/bash
# round 5, call g: the add for the brick after next asked for behind the scan (persistent waves)
OUT=gpurun_out/r5g; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
timeout 900 python tools/tail_ab.py --check --reps 5 --meshes torus1m,bunny16 --sets "auto:planheavy=0" > $OUT/tail_new_$rep.jsonl 2>> $OUT/err.log
DXV_LIBRARY=$D/libdxv_base.so timeout 600 python tools/tail_ab.py --reps 5 --meshes torus1m,bunny16 --sets "base:planregion=8,fuse=1" > $OUT/tail_base_$rep.jsonl 2>> $OUT/err.log
done
L=$D/libdxv_qtimes.so
for m in torus1m bunny16; do
  DXV_LIBRARY=$L QT_WORLD=8 timeout 300 python tools/queue_times.py $m 512 plan=2 >> $OUT/queue_times.jsonl 2>> $OUT/err.log
done
DXV_LIBRARY=$L timeout 300 python tools/queue_times.py torus1m 512 plan=2 >> $OUT/queue_times.jsonl 2>> $OUT/err.log
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5g/tail_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"], d["fresh"].get("queue_build_ms"))
for ln in open("gpurun_out/r5g/queue_times.jsonl"):
    d=json.loads(ln); print(d["mesh"], d["world"], d["kernel_ms"], d["mean_brick_us"], d["end_us_pct"], d["idle_wave_us_at_end_mean"])
PY
