#!/bin/bash
# round 4, call G: repeated launches through the queue ordered by measured chunk cost (queueorder) A/B, against round 3
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "work_queue or kept_memset or golden or headline_partition or config_grid" > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
for round in 1 2; do
echo "# old" >> $OUT/quick.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon >> $OUT/quick.jsonl 2>&1)
for o in 1 0; do
echo "# new queueorder=$o" >> $OUT/quick.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon --set queueorder=$o >> $OUT/quick.jsonl 2>&1
done; done
echo "# old" >> $OUT/quick256.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 >> $OUT/quick256.jsonl 2>&1)
for o in 1 0; do
echo "# new queueorder=$o" >> $OUT/quick256.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --set queueorder=$o >> $OUT/quick256.jsonl 2>&1
done
echo "# old" >> $OUT/rank.log
(cd .ab_old && timeout 300 python tools/rank_times.py torus1m 512 "lists=2" noparity zb8 | grep '"world": 8' >> $OUT/rank.log 2>&1)
for o in "lists=2,queueorder=1" "lists=2,queueorder=0"; do
echo "# $o" >> $OUT/rank.log
timeout 300 python tools/rank_times.py torus1m 512 "$o" noparity zb8 | grep '"world": 8' >> $OUT/rank.log 2>&1
timeout 300 python tools/rank_times.py bunny16 512 "$o" noparity zb8 | grep '"world": 8' >> $OUT/rank.log 2>&1
done
python - <<'PY'
import json
for f in ("quick.jsonl","quick256.jsonl"):
    for l in open('/root/repo/gpurun_out/r4g/'+f):
        if l.startswith('#'): print(l.strip()); continue
        try: d=json.loads(l)
        except Exception: print(l.strip()); continue
        print(d['mesh'], d['N'], 'queue', d['lists_ms'], 'bricks', d.get('plan_bricks'), d.get('lists_solid'))
PY
cat $OUT/rank.log
exit 0
