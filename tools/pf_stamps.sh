#!/bin/bash
# Diagnostic build of the library with s_memrealtime stamps in k_post_forward (mcts_kernels.hip) -> go-ai_amd/csrc/build/diag/libgoai_pf_stamps.so
# (extra hipcc flags as arguments); run on the GPU box as
#   GOAI_HIP_LIB=go-ai_amd/csrc/build/diag/libgoai_pf_stamps.so python tools/pf_stamps.py
set -e
cd "$(dirname "$0")/../go-ai_amd/csrc"
make -j8 >/dev/null
mkdir -p build/diag
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include"
OTHERS=$(ls build/*.o | grep -v mcts_kernels)
OUT=${OUT:-libgoai_pf_stamps.so}
/opt/rocm/bin/hipcc $FLAGS -DGOAI_PF_STAMPS "$@" -c mcts_kernels.hip -o build/diag/pf_stamps_$$.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/diag/$OUT $OTHERS build/diag/pf_stamps_$$.o
rm build/diag/pf_stamps_$$.o
ls -la build/diag/$OUT
