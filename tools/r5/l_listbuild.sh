#!/bin/bash
# round 5, call l: list build with the compact pair list and the 8-lane stop codes: same lists (tests), build times
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -k "lists or list or config or fuzz or frustum or refit or deferred or blob" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
python tools/build_once.py torus1m 5 > $OUT/build_torus1m.jsonl 2>&1; tail -2 $OUT/build_torus1m.jsonl
python tools/build_once.py bunny16 4 > $OUT/build_bunny16.jsonl 2>&1; tail -1 $OUT/build_bunny16.jsonl
python tools/build_once.py soup10m 3 > $OUT/build_soup10m.jsonl 2>&1; tail -2 $OUT/build_soup10m.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build_soup -- python3 $GRAFT_REPO_ROOT/tools/build_once.py soup10m 3 > $OUT/prof_build_soup.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof_build_soup -name "*kernel_stats.csv" | head -1); cp $f $OUT/prof_build_soup_kernel_stats.csv
find $OUT/prof_build_soup -name "*.csv" -size +1M -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r5m/prof_build_soup_kernel_stats.csv")))
for r in rows[:16]: print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
