#!/bin/bash
# experiment: waves per workgroup of the f32 row chain (T2S_ROWS_NW = 4 or 8)
for nw in "$@"; do
  make -C t2ms_amd/csrc clean > /dev/null
  make -C t2ms_amd/csrc FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DT2S_ROWS_NW=$nw" > /dev/null 2>&1 || { echo "build failed for $nw"; exit 1; }
  echo "== NW=$nw: $(python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],2), d["kernel_breakdown_us"])')"
done
make -C t2ms_amd/csrc clean > /dev/null
