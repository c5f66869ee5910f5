#!/bin/bash
# round 5, call k: where the build of the headline scene goes (LBVH + lists on the 512 map), kernel by kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_build_torus -- python3 $GRAFT_REPO_ROOT/tools/build_once.py torus1m 6 > $OUT/prof_build_torus.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof_build_torus -name "*kernel_stats.csv" | head -1); cp $f $OUT/prof_build_torus_kernel_stats.csv
find $OUT/prof_build_torus -name "*.csv" -size +4M -delete
tail -3 $OUT/prof_build_torus.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r5k/prof_build_torus_kernel_stats.csv")))
for r in rows[:32]: print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
