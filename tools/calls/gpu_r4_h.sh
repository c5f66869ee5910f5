#!/bin/bash
# round 4, call H: a brick's rays scan their lists together through LDS (coopscan) A/B
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4h
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "work_queue or kept_memset or golden or headline_partition or config_grid" > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
for round in 1 2; do
echo "# old" >> $OUT/quick.jsonl
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon >> $OUT/quick.jsonl 2>&1)
for o in 1 0; do
echo "# new coopscan=$o" >> $OUT/quick.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon,soup10m --set coopscan=$o >> $OUT/quick.jsonl 2>&1
done; done
for o in 1 0; do
echo "# new coopscan=$o" >> $OUT/quick256.jsonl
timeout 600 python tools/quick_times.py --meshes torus1m,bunny,dragon --grid 256 --set coopscan=$o >> $OUT/quick256.jsonl 2>&1
done
python - <<'PY'
import json
for f in ("quick.jsonl","quick256.jsonl"):
    for l in open('/root/repo/gpurun_out/r4h/'+f):
        if l.startswith('#'): print(l.strip()); continue
        try: d=json.loads(l)
        except Exception: print(l.strip()[:200]); continue
        print(d['mesh'], d['N'], 'queue', d['lists_ms'], 'bricks', d.get('plan_bricks'), d.get('lists_solid'))
PY
exit 0
