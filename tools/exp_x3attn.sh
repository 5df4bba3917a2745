#!/bin/bash
# timing ablations of the bf16x3 attention (results are garbage): T2S_EXP 131072 = no exp/split, 262144 = no MFMAs
for e in "$@"; do
  make -C t2ms_amd/csrc clean > /dev/null
  make -C t2ms_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DT2S_EXP=$e" > /dev/null 2>&1 || { echo "build failed for $e"; exit 1; }
  echo "== T2S_EXP=$e: $(python bench.py --steps 1 --warmup 1 --no-cpu-baseline --math bf16x3 --diffusion-steps 20 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_breakdown_us"])')"
done
make -C t2ms_amd/csrc clean > /dev/null
