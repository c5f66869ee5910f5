#!/bin/bash
# option share (the triangle step of a brick shared by its lanes): parity tests with it on, then timings off / on in one process each
OUT=gpurun_out/r4s; mkdir -p $OUT
DXV_OPTIONS="share=1" timeout 900 python -m pytest tests -m gpu -x -q -k "work_queue or grid_64 or texels or fuzz or synthetic or duplicates or largest or slabs or full_size or lists_equal" > $OUT/pytest_share.log 2>&1; tail -4 $OUT/pytest_share.log
for rep in 1 2; do
  for sh in 0 1; do
    DXV_OPTIONS="share=$sh" timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,soup10m --reps 9 --fresh > $OUT/q_share${sh}_$rep.jsonl 2>&1
  done
done
for sh in 0 1; do
  DXV_OPTIONS="share=$sh" timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_share$sh.json 2>$OUT/bench_share$sh.err
  DXV_OPTIONS="share=$sh" timeout 300 python tools/rank_times.py torus1m 512 lists=2 noparity zb8 > $OUT/rank_share$sh.jsonl 2>&1
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4s/q_*.jsonl")):
    lib=f.split("/q_")[1].rsplit("_",1)[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],lib)].append((d["lists_ms"],d.get("fresh_ms"),d["lists_solid"],d.get("queue_violations")))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob("gpurun_out/r4s/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["value"]), d["config"].get("solid_voxels"))
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r4s/rank_*.jsonl")):
    for ln in open(f):
        if '"world": 8' in ln: print(f.split("/")[-1], ln[150:300])
PY
