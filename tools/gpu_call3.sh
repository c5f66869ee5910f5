#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu3.log 2>&1
echo "# morton=1" > $OUT/sweep3.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 1,2,4,6 --stacks 0 --reps 5 --opts morton=1 >> $OUT/sweep3.log 2>&1
echo "# morton=0" >> $OUT/sweep3.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 1,2,4,6 --stacks 0 --reps 5 --opts morton=0 >> $OUT/sweep3.log 2>&1
echo "# 256 and parity" >> $OUT/sweep3.log
python tools/sweep.py --meshes torus1m --grids 256,512 --bricks 4 --stacks 0 --modes reference,parity --reps 5 >> $OUT/sweep3.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/tools/run_once.py
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc3_1 -- python3 $R torus1m 512 2 > $OUT/pmc3_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc3_2 -- python3 $R torus1m 512 2 > $OUT/pmc3_2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc3_5 -- python3 $R torus1m 512 2 > $OUT/pmc3_5.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc3_6 -- python3 $R torus1m 512 2 > $OUT/pmc3_6.log 2>&1
exit 0
