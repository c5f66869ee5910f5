#!/bin/bash
# round 4, call B: balance of the eight queues and what stealing is worth
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b
rm -rf $OUT; mkdir -p $OUT
for round in 1 2; do
for opt in "queuebalance=0" "queuebalance=1" "queuebalance=1,queuesteal=0"; do
echo "# $opt" >> $OUT/sweep.log
timeout 300 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,dragon --set $opt >> $OUT/sweep.log 2>&1
done; done
python - <<'PY'
import json
cur=None
for l in open('/root/repo/gpurun_out/r4b/sweep.log'):
    if l.startswith('#'): cur=l.strip(); continue
    try: d=json.loads(l)
    except Exception: print(l.strip()); continue
    print(cur, d['mesh'], d['lists_ms'], d.get('lists_solid'))
PY
exit 0
