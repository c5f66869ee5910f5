#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu2.log 2>&1
python tools/sweep.py --meshes torus1m --grids 512 --bricks 0,1,2,3,4,5,6,7 --stacks 0,12,32 --reps 3 > $OUT/sweep2.log 2>&1
python tools/sweep.py --meshes bunny,dragon --grids 512 --bricks 1,2,4,6 --stacks 0,12 --reps 3 >> $OUT/sweep2.log 2>&1
python bench.py --steps 10 --warmup 2 > $OUT/bench2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
R=$GRAFT_REPO_ROOT/tools/run_once.py
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc1 -- python3 $R torus1m 512 2 > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -- python3 $R torus1m 512 2 > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $R torus1m 512 2 > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $R torus1m 512 2 > $OUT/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc5 -- python3 $R torus1m 512 2 > $OUT/pmc5.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc6 -- python3 $R torus1m 512 2 > $OUT/pmc6.log 2>&1
exit 0
