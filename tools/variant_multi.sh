#!/bin/bash
# Build a variant of the library with SEVERAL sources of csrc/ compiled with the same extra -D flags:
#   tools/variant_multi.sh <name> "<stem> <stem> ..." <flags...>   ->  tools/bin/libt2s_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; stems=$2; shift 2
mkdir -p tools/bin
objs=""
for o in t2ms_amd/csrc/*.o; do
  st=$(basename $o .o); skip=0
  for s in $stems; do [ "$s" = "$st" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for s in $stems; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" \
      -c t2ms_amd/csrc/$s.hip -o tools/bin/${s}_$name.o
  objs="$objs tools/bin/${s}_$name.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libt2s_$name.so $objs
ls -la tools/bin/libt2s_$name.so
