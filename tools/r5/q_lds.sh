#!/bin/bash
for lds in 0 4096 6144; do for m in torus1m bunny16; do for n in 256 512; do
echo "lds $lds $m $n: $(DXV_LISTED_LDS=$lds python tools/r5/q_gap.py $m $n | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ms (median of 15, plan part)']; print('kept_hw', [x[0] for x in r['kept_hardware'][1:]])")"
done; done; done
