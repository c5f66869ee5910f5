#!/bin/bash
# round 5, call i: the suite, the smoke test and bench.py on the new defaults (nothing carried; Init builds the lists)
OUT=gpurun_out/r5i; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err; tail -3 $OUT/bench_driver_flags.err
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
python - <<'PY'
import json
for f in ("bench_driver_flags","bench"):
    try:
        d=json.loads(open(f"gpurun_out/r5i/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line", e); continue
    c=d["config"]
    print(f, "value", round(d["value"]), "ms", round(d["ms_per_step"],4), "roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if k not in ("note",)})
    print("  kept", c["kept_step"] and {k:(round(v,4) if isinstance(v,float) else v) for k,v in c["kept_step"].items() if k!="what"})
    print("  cold", c.get("first_voxelize_after_init"))
    print("  256", c.get("grid_256"), "frames2", c.get("frames_in_flight_2"), "frames3", c.get("frames_in_flight_3"))
    print("  cpu", d.get("cpu_baseline",{}).get("value"), "queue_build_ms", c["queue_build_ms"], "build", c["build_ms"], c["candidates"])
PY
