#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --interleave --no-cpu-baseline > $OUT/bench10_torchrun.log 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench10.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof10 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof10.log 2>&1
R=$GRAFT_REPO_ROOT/tools/run_once.py
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc10_f -- python3 $R torus1m 512 3 > $OUT/pmc10_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc10_w -- python3 $R torus1m 512 3 > $OUT/pmc10_w.log 2>&1
exit 0
