#!/bin/bash
# rocprofv3 evidence for one round, on the GPU box, in ONE gpurun call:
#   - bench.py --config c2..c5: one --kernel-trace --stats run each  -> <out>/<cfg>_kernel_stats.csv + bench line
#   - tools/bench_go.py (65,536 boards, 9x9 and 19x19): --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE passes
# Counter passes run on their own (never combined with a trace domain other than the kernel trace).
#   usage: tools/profile_round.sh <tag> [configs...]      output: gpurun_out/<tag>/
set -e
TAG=${1:-prof}; shift || true
CFGS=${*:-c2 c3 c4 c5}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $CFGS; do
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$cfg -- python3 $REPO/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 > $OUT/${cfg}_bench_line.json 2> $OUT/${cfg}_stats.err
    cp $(find $OUT/stats_$cfg -name '*kernel_stats.csv' | head -1) $OUT/${cfg}_kernel_stats.csv
    echo "$cfg done"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_go -- python3 $REPO/tools/bench_go.py --batch 65536 > $OUT/go_bench.txt 2> $OUT/go_stats.err
cp $(find $OUT/stats_go -name '*kernel_stats.csv' | head -1) $OUT/go_kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/go_fetch -- python3 $REPO/tools/bench_go.py --batch 65536 --iters 2 > $OUT/go_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/go_write -- python3 $REPO/tools/bench_go.py --batch 65536 --iters 2 > $OUT/go_write.log 2>&1
python3 $REPO/tools/pmc_go_summary.py $OUT > $OUT/go_pmc_summary.json
cat $OUT/go_bench.txt
cat $OUT/go_pmc_summary.json
