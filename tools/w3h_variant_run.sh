#!/bin/bash
# bench.py C2 (short) over variant libraries: tools/w3h_variant_run.sh <name>...   (on the GPU box; "base" = the shipped library)
cd "$(dirname "$0")/.."
for n in "$@"; do
    if [ "$n" = base ]; then unset GOAI_HIP_LIB; else export GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w3h_$n.so; fi
    timeout -k 10 150 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-exact-fp32 > gpurun_out/var_$n.json 2> gpurun_out/var_$n.err || exit 1
    python - "$n" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/var_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], flush=True)
PY
done
