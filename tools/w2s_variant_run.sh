#!/bin/bash
# bench.py --config <cfg> over conv_wino2s variant libraries: tools/w2s_variant_run.sh <cfg> <name>...  ("base" = the shipped library)
cd "$(dirname "$0")/.."
CFG=$1; shift
for n in "$@"; do
    if [ "$n" = base ]; then unset GOAI_HIP_LIB; else export GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w2s_$n.so; fi
    timeout -k 10 250 python bench.py --config $CFG --no-cpu-baseline --no-exact-fp32 > gpurun_out/var2_$n.json 2> gpurun_out/var2_$n.err || exit 1
    python - "$n" "$CFG" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/var2_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[1], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], flush=True)
PY
done
