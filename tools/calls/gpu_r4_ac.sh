#!/bin/bash
# hinted hardware dispatch of a queue built in the launch (size of the frame's previous queue) against persistent waves only (dispatch=0
# for rebuilt queues = previous behaviour is option dispatch 2 here: kept queues by hardware, fresh ones persistent)
OUT=gpurun_out/r4ac; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for rep in 1 2; do
  for d in 2 1; do
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny --reps 9 --fresh > $OUT/q_d${d}_$rep.jsonl 2>&1
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/quick_times.py --meshes torus1m,bunny --grid 256 --reps 9 --fresh > $OUT/q256_d${d}_$rep.jsonl 2>&1
    DXV_OPTIONS="dispatch=$d" timeout 300 python tools/refit_loop.py torus1m 512 40 > $OUT/refit_d${d}_$rep.jsonl 2>&1
  done
done
for d in 2 1; do DXV_OPTIONS="dispatch=$d" timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_d$d.json 2>$OUT/bench_d$d.err; done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4ac/q*_d*.jsonl")):
    tag=f.split("/")[-1].rsplit("_",1)[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],d["N"],tag[-2:])].append((d["lists_ms"],d.get("fresh_ms"),d["lists_solid"],d.get("fresh_solid")))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob("gpurun_out/r4ac/refit_*.jsonl")):
    for ln in open(f):
        if "not waited" in ln or '"device buffer"' in ln: d=json.loads(ln); print(f.split("/")[-1], d["vertices_from"][:30], d["fps"], d["list_ms"], d["voxelize_ms"])
for f in sorted(glob.glob("gpurun_out/r4ac/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); fs=d["config"].get("fresh_step") or {}; print(f.split("/")[-1], round(d["ms_per_step"],4), round(d["value"]), "fresh", fs.get("ms_per_step"), d["config"].get("solid_voxels"))
    except Exception as e: print(f, "ERR", e)
PY
