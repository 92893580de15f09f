#!/bin/bash
# PMC passes over the CNN forward (tools/bench_nn.py); each pass in its own rocprofv3 run.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_nn
mkdir -p $OUT
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/bench_nn.py --iters 3 > $OUT/$name.log 2>&1; }
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls $OUT/*/*
