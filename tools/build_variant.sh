#!/bin/bash
# A library build with extra compiler flags for kernels.hip, into build_ab/<name>.so (A/B inside one GPU call: tools/ab_lib.py)
#   bash tools/build_variant.sh pp4 -DTMX_SER_PP=4
set -e
name=$1; shift
cd "$(dirname "$0")/../tendermintx_amd/csrc"
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" -c kernels.hip -o /tmp/kernels_$name.o
g++ -shared -o ../../build_ab/$name.so /tmp/kernels_$name.o ntt.o trace.o poseidon.o value.o api.o codec.o
echo built build_ab/$name.so
