#!/bin/bash
# round 4, call F: helping out other heads only as an emergency measure: rank shares, full launches, wave timelines
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "work_queue or kept_memset or golden or headline_partition" > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
for round in 1 2; do
echo "# old" >> $OUT/quick.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon >> $OUT/quick.jsonl 2>&1)
echo "# new" >> $OUT/quick.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon --fresh >> $OUT/quick.jsonl 2>&1
done
echo "# old" >> $OUT/quick256.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 >> $OUT/quick256.jsonl 2>&1)
echo "# new" >> $OUT/quick256.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --fresh >> $OUT/quick256.jsonl 2>&1
for o in "lists=2" "lists=2,queuesteal=0"; do
echo "# $o" >> $OUT/rank.log
timeout 300 python tools/rank_times.py torus1m 512 "$o" noparity zb8 | grep '"world": 8' >> $OUT/rank.log 2>&1
timeout 300 python tools/rank_times.py bunny16 512 "$o" noparity zb8 | grep '"world": 8' >> $OUT/rank.log 2>&1
done
export DXV_LIBRARY=$GRAFT_REPO_ROOT/dxrvoxelizer_amd/libdxv_qtimes.so
QT_WORLD=8 timeout 300 python tools/queue_times.py torus1m 512 queuewaves=6144 >> $OUT/times.log 2>&1
timeout 300 python tools/queue_times.py torus1m 512 queuewaves=6144 >> $OUT/times.log 2>&1
unset DXV_LIBRARY
python - <<'PY'
import json
for f in ("quick.jsonl","quick256.jsonl"):
    for l in open('/root/repo/gpurun_out/r4f/'+f):
        if l.startswith('#'): print(l.strip()); continue
        try: d=json.loads(l)
        except Exception: print(l.strip()); continue
        print(d['mesh'], d['N'], 'queue', d['lists_ms'], 'fresh', d.get('fresh_ms'), 'bricks', d.get('plan_bricks'), 'viol', d.get('queue_violations'))
for l in open('/root/repo/gpurun_out/r4f/times.log'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['world'], d['kernel_ms'], 'busy', d['busy_frac_of_resident'], 'end', d['end_us_pct'], 'last', d['last_brick_us_pct']['50'], 'idle', d['idle_wave_us_at_end_mean'])
PY
cat $OUT/rank.log
exit 0
