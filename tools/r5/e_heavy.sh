#!/bin/bash
# round 5, call e: heavy bricks first (two-ended queues) -- thresholds against "no brick is heavy", same library, same box; time lines
OUT=gpurun_out/r5e; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python tools/tail_ab.py --check --reps 7 --sets "off:planheavy=65535;h48:planheavy=48;h24:planheavy=24;h16:planheavy=16;h12:planheavy=12" > $OUT/tail_heavy.jsonl 2>> $OUT/err.log
DXV_LIBRARY=$D/libdxv_base.so timeout 600 python tools/tail_ab.py --reps 7 --sets "base:planregion=8,fuse=1" > $OUT/tail_base.jsonl 2>> $OUT/err.log
L=$D/libdxv_qtimes.so
for m in torus1m bunny16; do
  DXV_LIBRARY=$L timeout 300 python tools/wg_times.py $m 512 8 4 planheavy=24 >> $OUT/wg_times.jsonl 2>> $OUT/err.log
done
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in ("gpurun_out/r5e/tail_heavy.jsonl","gpurun_out/r5e/tail_base.jsonl"):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"])
for ln in open("gpurun_out/r5e/wg_times.jsonl"):
    d=json.loads(ln)
    print(d["mesh"], d["world"], d["rank"], d["kernel_ms_events"], d["span_us"], d["ideal_us_at_peak_concurrency"], d["us_below_50pct_of_peak_at_end"], d["in_flight_over_time_40_bins"][26:], d["mean_brick_us_by_start_decile"])
PY
