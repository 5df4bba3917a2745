#!/bin/bash
# Upper bound of what a register-resident bf16 FORWARD chain per half block (stage A) can save, MEASURED
# before building it: tools/bin/libt2s_chain_bound.so is the in-tree library with -DT2S_CHAIN_BOUND (csrc/t2s_bf16.h: the forward
# GEMMs skip exactly the re-reads the chain removes; results invalid).  Build here (hipcc cross-compiles), then on the GPU box:
#   tools/chain_bound.sh build          # in the build container
#   tools/chain_bound.sh run [rounds]   # on the GPU box: same-box A/B with tools/ab_lib.sh
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DT2S_CHAIN_BOUND \
      -c t2ms_amd/csrc/t2s_train.hip -o tools/bin/t2s_train_chain_bound.o
  objs=$(ls t2ms_amd/csrc/*.o | grep -v t2s_train.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libt2s_chain_bound.so $objs tools/bin/t2s_train_chain_bound.o
  ls -la tools/bin/libt2s_chain_bound.so
else
  bash tools/ab_lib.sh tools/bin/libt2s_chain_bound.so "${2:-3}"
fi
