# runs tools/wino3h_stamps.py on the diagnostic libraries named as arguments (go-ai_amd/csrc/build/diag/libgoai_w3h_<name>.so)
for d in "$@"; do echo "== $d"; GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w3h_$d.so timeout -k 10 100 python tools/wino3h_stamps.py > gpurun_out/w3h_$d.txt 2>&1 || exit 1; grep -E "forward|epilogue  |layer  |stage =" gpurun_out/w3h_$d.txt; done
