#!/bin/bash
# per-kernel times of Init (LBVH + lists) for a small and the headline mesh
OUT=gpurun_out/r5q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for m in ${MESHES:-bunny torus1m}; do
  rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $GRAFT_REPO_ROOT/tools/build_once.py $m 6 > /tmp/pb.log 2>&1
  f=$(find /tmp/pb -name "*kernel_stats.csv" | head -1); echo "== $m"; tail -2 /tmp/pb.log | cut -c1-300
  python3 -c "
import csv
for r in csv.DictReader(open('$f')): print('   %-46s calls %4s avg_us %8.1f'%(r['Name'].replace('dxv::','').replace('(anonymous namespace)::','')[:46], r['Calls'], float(r['AverageNs'])/1e3))
"
done > $GRAFT_REPO_ROOT/$OUT/prof_build.txt 2>&1
cat $GRAFT_REPO_ROOT/$OUT/prof_build.txt
