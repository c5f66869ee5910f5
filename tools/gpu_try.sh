#!/bin/bash
# one quick look at new code on the GPU: a few test files, then a short bench; logs under gpurun_out/try
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/try
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest ${TRY_TESTS:-tests/test_gpu_prepared.py} -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1
tail -25 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -3 $OUT/smoke.log
timeout 600 python bench.py ${TRY_BENCH:---no-cpu-baseline} > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/try/bench.json")
try:
    d=json.load(open(p)); c=d["config"]
    print("value",d["value"],"ms",d["ms_per_step"],"kernel",d["roofline"]["kernel"],d["roofline"]["kernel_ms"])
    for k in ("unprepared_step","kept_step"):
        if c.get(k): print(k,c[k]["ms_per_step"],c[k]["mvoxels_s"])
    print("256",c.get("grid_256")); print("inflight",c.get("frames_in_flight_2"),c.get("frames_in_flight_3")); print("cold",c.get("first_voxelize_after_init"))
    print("prepare_ms",c.get("queue_prepare_ms"),"warmup",c.get("warmup_ms"))
except Exception as e: print("no bench line",e)
PY
exit 0
