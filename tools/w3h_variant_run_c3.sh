#!/bin/bash
# as w3h_variant_run.sh, bench.py --config c3
cd "$(dirname "$0")/.."
for n in "$@"; do
    if [ "$n" = base ]; then unset GOAI_HIP_LIB; else export GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w3h_$n.so; fi
    timeout -k 10 150 python bench.py --config c3 --no-cpu-baseline --no-exact-fp32 > gpurun_out/var3_$n.json 2> gpurun_out/var3_$n.err || exit 1
    python - "$n" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/var3_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], flush=True)
PY
done
