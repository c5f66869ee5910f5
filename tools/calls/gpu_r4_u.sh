#!/bin/bash
# option dispatch: a kept queue whose lengths the host knows dealt out by the hardware (one workgroup per brick) against persistent waves
OUT=gpurun_out/r4u; mkdir -p $OUT
DXV_OPTIONS="dispatch=1" timeout 900 python -m pytest tests -m gpu -x -q -k "work_queue or grid_64 or texels or slabs or interleaved or frames or kept_memset" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for rep in 1 2; do
  for d in 0 1; do
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --reps 15 > $OUT/q256_d${d}_$rep.jsonl 2>&1
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 15 > $OUT/q512_d${d}_$rep.jsonl 2>&1
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_d${d}_$rep.jsonl 2>&1
  done
done
for d in 0 1; do DXV_OPTIONS="dispatch=$d" timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_d$d.json 2>$OUT/bench_d$d.err; done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4u/q*_d*.jsonl")):
    tag=f.split("/")[-1].rsplit("_",1)[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],tag)].append((d["lists_ms"],d["plan_waves"],d["lists_solid"]))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob("gpurun_out/r4u/rank_*.jsonl")):
    for ln in open(f):
        if '"world"' in ln: d=json.loads(ln); print(f.split("/")[-1], d["world"], d["rank_ms"], d["ideal_speedup"])
for f in sorted(glob.glob("gpurun_out/r4u/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), d["config"].get("solid_voxels"), d["config"].get("work_queue"))
    except Exception as e: print(f, "ERR", e)
PY
