#!/bin/bash
# round 5, call f: the scene's own "long list" threshold + the next brick's word fetched during the current brick (persistent waves)
OUT=gpurun_out/r5f; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python tools/tail_ab.py --check --reps 7 --meshes torus1m,bunny16,dragon9 --sets "off:planheavy=65535;auto:planheavy=0;h16:planheavy=16;auto6:planheavy=0,planregion=6" > $OUT/tail_auto.jsonl 2>> $OUT/err.log
DXV_LIBRARY=$D/libdxv_base.so timeout 600 python tools/tail_ab.py --reps 7 --meshes torus1m,bunny16,dragon9 --sets "base:planregion=8,fuse=1" > $OUT/tail_base.jsonl 2>> $OUT/err.log
L=$D/libdxv_qtimes.so
for m in torus1m bunny16; do
  DXV_LIBRARY=$L QT_WORLD=8 timeout 300 python tools/queue_times.py $m 512 plan=2 >> $OUT/queue_times.jsonl 2>> $OUT/err.log
done
tail -5 $OUT/err.log
python - <<'PY'
import json,glob
for f in ("gpurun_out/r5f/tail_auto.jsonl","gpurun_out/r5f/tail_base.jsonl"):
    for ln in open(f):
        d=json.loads(ln)
        print(d["mesh"], d["set"], "kept", d["kept"]["full_ms"], d["kept"]["slowest_rank_ms"], d["kept"]["ideal_speedup"], d["kept"]["g256_ms"], "| fresh", d["fresh"]["full_ms"], d["fresh"]["slowest_rank_ms"], d["fresh"]["ideal_speedup"], d["fresh"]["g256_ms"], d["fresh"].get("queue_build_ms"))
for ln in open("gpurun_out/r5f/queue_times.jsonl"): print(ln[:1500])
PY
