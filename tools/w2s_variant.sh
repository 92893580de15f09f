#!/bin/bash
# Variant builds of the library with extra flags on conv_wino2s_kernels.hip -> go-ai_amd/csrc/build/diag/libgoai_w2s_<name>.so
#   usage: tools/w2s_variant.sh <name> [hipcc flags...]
set -e
NAME=$1; shift
cd "$(dirname "$0")/../go-ai_amd/csrc"
mkdir -p build/diag
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -mllvm -pragma-unroll-threshold=200000"
OTHERS=$(ls build/*.o | grep -v conv_wino2s_kernels)
/opt/rocm/bin/hipcc $FLAGS "$@" -c conv_wino2s_kernels.hip -o build/diag/w2s_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/diag/libgoai_w2s_$NAME.so $OTHERS build/diag/w2s_$NAME.o
rm build/diag/w2s_$NAME.o
