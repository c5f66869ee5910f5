#!/bin/bash
OUT=gpurun_out/r5q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in time1 time6; do for plan in 0 24; do
  rm -rf /tmp/sp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -- $GRAFT_REPO_ROOT/tools/micro/sort_check $c $plan > /dev/null 2>&1
  f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1); echo "$c plan $plan"
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')): print('   %-40s calls %4s avg_us %8.1f'%(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3))
"
done; done > $GRAFT_REPO_ROOT/$OUT/sort_prof.txt 2>&1
cat $GRAFT_REPO_ROOT/$OUT/sort_prof.txt
