#!/bin/bash
# Same-box comparison of the in-tree library and any number of variant builds on the training step:
#   tools/ab_libs.sh [-r rounds] [VAR=VALUE ...] <lib.so ...>     (VAR=VALUE entries run the in-tree library with that environment)
rounds=2
if [ "$1" = -r ]; then rounds=$2; shift 2; fi
for r in $(seq $rounds); do
  for what in "" "$@"; do
    unset T2S_LIB; envs=""
    case "$what" in
      "") tagname="in-tree";;
      *=*) envs="$what"; tagname="$what";;
      *) export T2S_LIB=$what; tagname="$what";;
    esac
    env $envs python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_classes']
print('$tagname'.ljust(40), round(t['ms_per_step'],3), 'ms', {a[6:]: round(v['ms_per_step'],2) for a,v in k.items()})"
  done
done
