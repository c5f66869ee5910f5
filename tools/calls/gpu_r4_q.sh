#!/bin/bash
# A/B on one box: the next brick's texel words sent into the LDS ahead of time (global_load_lds) against the committed kernel
OUT=gpurun_out/r4q; mkdir -p $OUT
V=${VARIANT:-glds}
D=$PWD/dxrvoxelizer_amd
DXV_LIBRARY=$D/libdxv_$V.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "work_queue or grid_64 or texels" > $OUT/pytest_$V.log 2>&1; tail -3 $OUT/pytest_$V.log
for rep in 1 2 3; do
  for lib in libdxv.so libdxv_$V.so; do
    DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 15 --fresh > $OUT/q_${lib}_$rep.jsonl 2>&1
  done
done
for lib in libdxv.so libdxv_$V.so; do
  DXV_LIBRARY=$D/$lib timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $OUT/bench_$lib.json 2>$OUT/bench_$lib.err
  DXV_LIBRARY=$D/$lib timeout 300 python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_$lib.jsonl 2>&1
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4q/q_*.jsonl")):
    lib=f.split("q_")[1].rsplit("_",1)[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],lib)].append((d["lists_ms"],d.get("fresh_ms"),d["lists_solid"],d.get("queue_violations")))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob("gpurun_out/r4q/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), d["config"].get("fresh_step",{}).get("ms_per_step"))
for f in sorted(glob.glob("gpurun_out/r4q/rank_*.jsonl")):
    for ln in open(f):
        if '"world": 8' in ln: print(f, ln[:300])
PY
