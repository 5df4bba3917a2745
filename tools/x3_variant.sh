#!/bin/bash
# Build a variant of the library whose t2s_dit.hip (the bf16x3 row kernels live there) is compiled with extra -D flags:
#   tools/x3_variant.sh <name> <flags...>   ->  tools/bin/libt2s_<name>.so      (A/B: T2S_LIB=tools/bin/libt2s_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" \
    -c t2ms_amd/csrc/t2s_dit.hip -o tools/bin/t2s_dit_$name.o
objs=$(ls t2ms_amd/csrc/*.o | grep -v t2s_dit.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libt2s_$name.so $objs tools/bin/t2s_dit_$name.o
ls -la tools/bin/libt2s_$name.so
