#!/bin/bash
# radial range of an entry cut to its texel (dm_local_radial) against the whole footprint's range: parity and exhaustive list checks
# with it, then timings (kernel and list build) on one box, libraries alternating
OUT=gpurun_out/r4v; mkdir -p $OUT
D=$PWD/dxrvoxelizer_amd
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
[ -n "$SKIP_CHECK" ] || timeout 900 python tools/list_check_configs.py > $OUT/list_check_configs.jsonl 2>&1; tail -3 $OUT/list_check_configs.jsonl | cut -c1-200
for rep in 1 2; do
  for lib in libdxv_base.so libdxv.so; do
    DXV_LIBRARY=$D/$lib timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m --reps 9 --fresh > $OUT/q_${lib}_$rep.jsonl 2>&1
  done
done
for lib in libdxv_base.so libdxv.so; do
  DXV_LIBRARY=$D/$lib timeout 300 python tools/quick_times.py --meshes bunny,dragon9 --grid 1024 --reps 3 > $OUT/q1024_$lib.jsonl 2>&1
  DXV_LIBRARY=$D/$lib timeout 300 python tools/refit_loop.py torus1m 512 40 > $OUT/refit_$lib.jsonl 2>&1
  DXV_LIBRARY=$D/$lib timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_$lib.json 2>$OUT/bench_$lib.err
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r4v/q_*.jsonl")+glob.glob("gpurun_out/r4v/q1024_*.jsonl")):
    lib=f.split("/")[-1].split("_lib")[1].split(".so")[0]
    for ln in open(f):
        if ln.startswith("{"):
            d=json.loads(ln); acc[(d["mesh"],d["N"],lib)].append((d["lists_ms"],d.get("fresh_ms"),d["list_ms"],d["entries"],d["lists_solid"]))
for k,v in sorted(acc.items()): print(k, v)
for f in sorted(glob.glob("gpurun_out/r4v/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"],4), round(d["value"]), d["config"].get("solid_voxels"))
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r4v/refit_*.jsonl")):
    for ln in open(f):
        if "not waited" in ln or '"device buffer"' in ln: d=json.loads(ln); print(f.split("/")[-1], d["vertices_from"][:30], d["fps"], d["list_ms"], d["voxelize_ms"])
PY
