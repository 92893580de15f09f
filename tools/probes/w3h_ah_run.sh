for a in 1 2 3; do echo "== AH $a"; GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w3h_ah$a.so timeout -k 10 150 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-exact-fp32 > gpurun_out/ah$a.json || exit 1; python -c "
import json
d=json.load(open('gpurun_out/ah$a.json')); print(d['value'], d['roofline']['avg_launch_us'])"; done
