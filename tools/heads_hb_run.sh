#!/bin/bash
# bench.py --config <cfg> with k_heads over two / eight boards per workgroup
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  for hb in 2 8 2 8; do
    GOAI_HEADS_HB=$hb timeout -k 10 250 python bench.py --config $cfg --no-cpu-baseline --no-exact-fp32 > gpurun_out/hb_${cfg}_$hb.json 2> gpurun_out/hb_${cfg}_$hb.err || exit 1
    python - "$cfg" "$hb" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/hb_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "HB", sys.argv[2], d["value"], d["ms_per_step"], flush=True)
PY
  done
done
