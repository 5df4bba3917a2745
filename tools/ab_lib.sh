#!/bin/bash
# A/B two builds of the library on ONE box (box-to-box spread is ~1.5-5 %, larger than most kernel changes):
#   tools/ab_lib.sh <other .so> [rounds]    alternates tools/bench_train.py between the in-tree build and T2S_LIB=<other>
other=${1:-t2ms_amd/libt2s_hip_prev.so}; rounds=${2:-3}
for r in $(seq $rounds); do
  for lib in "" "$other"; do
    if [ -n "$lib" ]; then export T2S_LIB=$lib; else unset T2S_LIB; fi
    python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_classes']
print('${lib:-in-tree}'.ljust(34), round(t['ms_per_step'],3), 'ms', {a[6:]: round(v['ms_per_step'],2) for a,v in k.items()})"
  done
done
