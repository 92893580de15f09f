#!/bin/bash
# Diagnostic build of the library with s_memrealtime stamps in conv_wino2s_kernels.hip -> go-ai_amd/csrc/build/diag/libgoai_w2s_stamps.so
# (extra hipcc flags as arguments); run on the GPU box as
#   GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_w2s_stamps.so python tools/wino2s_stamps.py
set -e
cd "$(dirname "$0")/../go-ai_amd/csrc"
make -j8 >/dev/null
mkdir -p build/diag
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -mllvm -pragma-unroll-threshold=200000"
OTHERS=$(ls build/*.o | grep -v conv_wino2s_kernels)
OUT=${OUT:-libgoai_w2s_stamps.so}
/opt/rocm/bin/hipcc $FLAGS -DGOAI_CONV_STAMPS "$@" -c conv_wino2s_kernels.hip -o build/diag/w2s_stamps_$$.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/diag/$OUT $OTHERS build/diag/w2s_stamps_$$.o
rm build/diag/w2s_stamps_$$.o
ls -la build/diag/$OUT
