#!/bin/bash
for cp in "$@"; do
  make -C t2ms_amd/csrc clean > /dev/null
  make -C t2ms_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DT2S_VAE_CP=$cp" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "== CP=$cp: $(python tools/bench_train.py --batch 1152 --steps 3 --warmup 1 --dtype bf16 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
done
make -C t2ms_amd/csrc clean > /dev/null
