#!/bin/bash
# round 5, call c: equal shares of the eight queues (queue_item) against the build before it, same box, alternating
OUT=gpurun_out/r5c; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
  for lib in libdxv_base.so libdxv.so; do
    DXV_LIBRARY=$D/$lib timeout 600 python tools/tail_ab.py --check --reps 5 --sets "auto:planregion=0,fuse=1;r8:planregion=8,fuse=1" > $OUT/tail_${lib}_$rep.jsonl 2>> $OUT/err.log
  done
done
L=$D/libdxv_qtimes.so
for m in torus1m bunny16; do
  DXV_LIBRARY=$L timeout 300 python tools/wg_times.py $m 512 8 4 planregion=8 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
done
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c/tail_*.jsonl")):
    for ln in open(f):
        d=json.loads(ln)
        print(f.split("/")[-1][5:-6], d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"])
for ln in open("gpurun_out/r5c/wg_times.jsonl"):
    d=json.loads(ln); print(d["mesh"], d["rank"], d["kernel_ms_events"], d["span_us"], d["ideal_us_at_peak_concurrency"], d["end_by_xcc_us"], d["bricks_by_xcc"], d["us_below_50pct_of_peak_at_end"])
PY
