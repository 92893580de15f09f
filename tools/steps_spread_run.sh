#!/bin/bash
# bench.py C2 with k_search_steps over start spreads (GOAI_STEPS_SPREAD, 64-clock units) and without the kernel ("off")
cd "$(dirname "$0")/.."
for n in "$@"; do
    unset GOAI_STEPS_KERNEL GOAI_STEPS_SPREAD
    if [ "$n" != off ]; then export GOAI_STEPS_KERNEL=1; fi
    if [ "$n" != off ] && [ "$n" != default ]; then export GOAI_STEPS_SPREAD=$n; fi
    timeout -k 10 150 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-exact-fp32 > gpurun_out/sp_$n.json 2> gpurun_out/sp_$n.err || exit 1
    python - "$n" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/sp_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], flush=True)
PY
done
