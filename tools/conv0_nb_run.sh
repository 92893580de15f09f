#!/bin/bash
# bench.py --config <cfg> with k_conv0_mfma over two / eight boards per workgroup
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  for nb in 2 8 2 8; do
    GOAI_CONV0_NB=$nb timeout -k 10 250 python bench.py --config $cfg --no-cpu-baseline --no-exact-fp32 > gpurun_out/nb_${cfg}_$nb.json 2> gpurun_out/nb_${cfg}_$nb.err || exit 1
    python - "$cfg" "$nb" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/nb_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], "NB", sys.argv[2], d["value"], d["ms_per_step"], flush=True)
PY
  done
done
