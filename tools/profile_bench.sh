#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: one --kernel-trace --stats run, then PMC passes,
# each in a run of its own (counter blocks are not split over passes).  Output: gpurun_out/<tag>/.
#   usage: tools/profile_bench.sh <tag> [bench.py args...]
set -e
TAG=${1:-prof}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
pmc() { name=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-exact-fp32 ${EXTRA} > $OUT/$name.log 2>&1; }
EXTRA="$*"
pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $REPO/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
