#!/bin/bash
# experiment: LLVM AMDGPU scheduling strategy for one translation unit (per-file FLAGS_<stem> of the Makefile)
#   tools/exp_sched.sh t2s_attn_x3 max-ilp max-memory-clause ...
stem=$1; shift
for st in "$@"; do
  make -C t2ms_amd/csrc clean > /dev/null
  if [ "$st" = "default" ]; then extra=""; else extra="-mllvm -amdgpu-sched-strategy=$st"; fi
  make -C t2ms_amd/csrc "FLAGS_$stem=$extra" > /dev/null 2>&1 || { echo "build failed for $st"; exit 1; }
  echo "== $stem $st: f32 $(python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --diffusion-steps 100 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_breakdown_us"])')"
  echo "   x3 $(python bench.py --steps 1 --warmup 1 --no-cpu-baseline --math bf16x3 --diffusion-steps 100 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["kernel_breakdown_us"])')"
done
make -C t2ms_amd/csrc clean > /dev/null
