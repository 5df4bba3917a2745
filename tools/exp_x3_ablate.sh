#!/bin/bash
# Timing ablations of the bf16x3 attention kernel (results invalid): builds with -DT2S_ABL=<bits> for each argument
# (see t2s_attn_x3.hip) and prints the in-situ kernel times.  Run on the GPU box.
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
for e in "$@"; do
  make -C t2ms_amd/csrc clean > /dev/null
  make -C t2ms_amd/csrc FLAGS="$F -DT2S_ABL=$e" > /dev/null 2>&1 || { echo "build failed for $e"; exit 1; }
  echo "T2S_ABL=$e: $(timeout -k 10 120 python tools/time_x3_insitu.py 2>/dev/null | tail -1)"
done
make -C t2ms_amd/csrc clean > /dev/null
make -C t2ms_amd/csrc > /dev/null 2>&1
