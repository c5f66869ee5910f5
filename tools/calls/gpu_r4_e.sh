#!/bin/bash
# round 4, call E: all 64 heads probed at once when a head is done: full launches, rank shares, wave / brick times
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "work_queue or kept_memset or golden or headline_partition" > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
for round in 1 2; do
echo "# old round $round" >> $OUT/quick.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon >> $OUT/quick.jsonl 2>&1)
echo "# new round $round" >> $OUT/quick.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon --fresh >> $OUT/quick.jsonl 2>&1
done
echo "# old" >> $OUT/quick256.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 >> $OUT/quick256.jsonl 2>&1)
echo "# new" >> $OUT/quick256.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --fresh >> $OUT/quick256.jsonl 2>&1
echo "# old" >> $OUT/rank_times.jsonl
(cd .ab_old && timeout 600 python tools/rank_times.py torus1m 512 "lists=2" noparity zb8 >> $OUT/rank_times.jsonl 2>&1)
echo "# new" >> $OUT/rank_times.jsonl
timeout 600 python tools/rank_times.py torus1m 512 "lists=2" noparity zb8 >> $OUT/rank_times.jsonl 2>&1
timeout 600 python tools/rank_times.py bunny16 512 "lists=2" noparity zb8 >> $OUT/rank_times.jsonl 2>&1
for n in 512 256; do
DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so timeout 300 python tools/queue_times.py torus1m $n >> $OUT/times.log 2>&1
done
python - <<'PY'
import json
for f in ("quick.jsonl","quick256.jsonl"):
    for l in open('/root/repo/gpurun_out/r4e/'+f):
        if l.startswith('#'): print(l.strip()); continue
        try: d=json.loads(l)
        except Exception: print(l.strip()); continue
        print(d['mesh'], d['N'], 'queue', d['lists_ms'], 'fresh', d.get('fresh_ms'), 'plan', d.get('plan_ms'), 'bricks', d.get('plan_bricks'), 'viol', d.get('queue_violations'))
PY
grep '"world": 8\|^#' $OUT/rank_times.jsonl; cat $OUT/times.log
exit 0
