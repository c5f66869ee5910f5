#!/bin/bash
# round 4, call A: work queue + persistent waves -- GPU tests of the queue, then old (round 3: host-built dispatch plan) against new
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "work_queue or kept_memset or golden or smoke or frames_in_flight" > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
for round in 1 2; do
echo "# old round $round" >> $OUT/ab.log
(cd .ab_old && timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon >> $OUT/ab.log 2>&1)
echo "# new round $round" >> $OUT/ab.log
timeout 600 python tools/quick_times.py --meshes torus1m,bunny16,dragon9,bunny,dragon --fresh >> $OUT/ab.log 2>&1
done
cat $OUT/ab.log
exit 0
