#!/bin/bash
# A/B: bf16x3 kernels with packed fp32 VALU ops (T2S_EXP=2048) vs without (default); run on the GPU box.
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
for e in 2048 0; do
  make -C t2ms_amd/csrc clean > /dev/null
  make -C t2ms_amd/csrc FLAGS="$F -DT2S_EXP=$e" > /dev/null 2>&1 || { echo "build failed for $e"; exit 1; }
  for l in 1 2; do
    echo "T2S_EXP=$e lanes=$l: $(timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --math bf16x3 --lanes $l --diffusion-steps 300 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]/300,4), "ms/step", d["kernel_breakdown_us"])')"
  done
done
make -C t2ms_amd/csrc clean > /dev/null
make -C t2ms_amd/csrc > /dev/null 2>&1
