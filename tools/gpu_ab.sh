#!/bin/bash
# same-box A/B of two builds of the library: tools/gpu_ab.sh TAG [other library name, default libdxv_base.so]
TAG=${1:-ab}; OTHER=${2:-libdxv_base.so}
OUT=gpurun_out/$TAG; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 900 python -m pytest tests -m gpu -x -q -k "work_queue or grid_64 or texels or fuzz or synthetic or kernel_variant or largest or slabs" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2 3; do
  for lib in $OTHER libdxv.so; do
    DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 15 --fresh > $OUT/q_${lib}_$rep.jsonl 2>&1
  done
done
for lib in $OTHER libdxv.so; do
  DXV_LIBRARY=$D/$lib timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > $OUT/bench_$lib.json 2>$OUT/bench_$lib.err
  DXV_LIBRARY=$D/$lib timeout 300 python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_$lib.jsonl 2>&1
  DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes torus1m,dragon9 --grid 1024 --reps 5 > $OUT/q1024_$lib.jsonl 2>&1
  DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes torus1m,bunny --grid 256 --reps 9 > $OUT/q256_$lib.jsonl 2>&1
done
python - "$OUT" <<'PY'
import json,glob,collections,sys
out=sys.argv[1]
acc=collections.defaultdict(list)
for f in sorted(glob.glob(out+"/q_*.jsonl")):
    lib=f.split("/q_")[1].rsplit("_",1)[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],lib)].append((d["lists_ms"],d.get("fresh_ms"),d["lists_solid"],d.get("queue_violations")))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob(out+"/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), d["config"].get("fresh_step",{}).get("ms_per_step"))
for f in sorted(glob.glob(out+"/rank_*.jsonl")):
    for ln in open(f):
        if '"world": 8' in ln: print(f.split("/")[-1], ln[150:300])
for f in sorted(glob.glob(out+"/q1024_*.jsonl")+glob.glob(out+"/q256_*.jsonl")):
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); print(f.split("/")[-1], d["mesh"], d["N"], d["lists_ms"], d["lists_solid"])
PY
