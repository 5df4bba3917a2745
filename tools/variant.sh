#!/bin/bash
# Build a variant of the library with ONE source of csrc/ compiled with extra -D flags (diagnosis builds; results may be invalid):
#   tools/variant.sh <stem> <name> <flags...>   ->  tools/bin/libt2s_<name>.so      (A/B: T2S_LIB=tools/bin/libt2s_<name>.so)
set -e
cd "$(dirname "$0")/.."
stem=$1; name=$2; shift 2
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" \
    -c t2ms_amd/csrc/$stem.hip -o tools/bin/${stem}_$name.o
objs=$(ls t2ms_amd/csrc/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libt2s_$name.so $objs tools/bin/${stem}_$name.o
ls -la tools/bin/libt2s_$name.so
