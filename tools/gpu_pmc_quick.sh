#!/bin/bash
# PMC counters of the voxelize kernel in separate passes (counter passes never combined with trace domains other than
# --kernel-trace): tools/gpu_pmc_quick.sh TAG [mesh N key=value ...]  ->  gpurun_out/pmcq/TAG/*.csv + summary.json
TAG=${1:-run}; shift
MESH=${1:-torus1m}; shift
N=${1:-512}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (no warm-up pass in dxv_create: its one small launch of the same kernel would be averaged into the per-launch figures)
export DXV_WARMUP=0
R=$GRAFT_REPO_ROOT/tools/run_once.py
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R $MESH $N ${PMC_LAUNCHES:-4} reference lists=2 $EXTRA > $OUT/$name.log 2>&1; }
EXTRA="$*"
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_WR
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
# (queued bricks of the launch, for the per-brick figures of the persistent kernel: from the driver's own stats line)
export PMC_BRICKS=$(grep -o "'plan_bricks': [0-9]*" $OUT/sq1.log | tail -1 | grep -o "[0-9]*$")
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT
