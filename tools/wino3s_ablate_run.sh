#!/bin/bash
# On the GPU box: the stamp report of every diagnostic build made by tools/wino3s_stamps.sh -DW3S_DIAG=<mask>
cd "$(dirname "$0")/.."
for m in ${MASKS:-0 1 2 4 8 16 3 7}; do
  echo "=== W3S_DIAG=$m"
  GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w3s_diag_$m.so timeout -k 10 120 python tools/wino3s_stamps.py 2>&1 | grep -E "forward|layer  |a stage|epilogue|image staged" || exit 1
done
