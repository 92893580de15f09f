#!/bin/bash
# Diagnostic: builds libgoai_hip variants whose F(3x3,3x3) loop leaves parts out (W3_DIAG bit mask, see
# conv_wino3_kernels.hip) into go-ai_amd/csrc/build/diag/, to be timed on the GPU box in ONE call:
#   for m in $MASKS; do GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_diag_$m.so python tools/bench_nn.py; done
set -e
cd "$(dirname "$0")/../go-ai_amd/csrc"
make -j8 >/dev/null
mkdir -p build/diag
MASKS=${MASKS:-"0 1 2 4 8 16 3 7 15 31 32 63"}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include"
OTHERS=$(ls build/*.o | grep -v conv_wino3_kernels)
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc $FLAGS -DW3_DIAG=$m -c conv_wino3_kernels.hip -o build/diag/w3_$m.o 2>/dev/null &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/diag/libgoai_diag_$m.so $OTHERS build/diag/w3_$m.o ) &
  while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 1; done
done
wait
ls -la build/diag/*.so
