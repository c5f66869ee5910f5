#!/bin/bash
OUT=gpurun_out/r5q; mkdir -p $OUT
DXV_ALLOW_API_MISMATCH=1 timeout 600 python tools/ablate.py --meshes torus1m,bunny16 --values 0,8,1,2,4,32,64,16 > $OUT/ablate.jsonl 2> $OUT/err.log
tail -3 $OUT/err.log; cat $OUT/ablate.jsonl
