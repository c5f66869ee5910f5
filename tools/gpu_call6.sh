#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
(time python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider) > $OUT/pytest_gpu6.log 2>&1
: > $OUT/sweep6.log
for q in 0 1; do
echo "# queue=$q" >> $OUT/sweep6.log
python tools/sweep.py --meshes torus1m,bunny,dragon --grids 512 --bricks 4 --stacks 0 --reps 5 --opts queue=$q >> $OUT/sweep6.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/tools/run_once.py
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc6_1 -- python3 $R torus1m 512 2 > $OUT/pmc6_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS --output-format csv -d $OUT/pmc6_2 -- python3 $R torus1m 512 2 > $OUT/pmc6_2.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum --output-format csv -d $OUT/pmc6_3 -- python3 $R torus1m 512 2 > $OUT/pmc6_3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmc6_4 -- python3 $R torus1m 512 2 > $OUT/pmc6_4.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc6_5 -- python3 $R torus1m 512 2 > $OUT/pmc6_5.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc6_6 -- python3 $R torus1m 512 2 > $OUT/pmc6_6.log 2>&1
exit 0
