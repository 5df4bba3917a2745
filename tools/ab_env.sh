#!/bin/bash
# A/B an environment switch on ONE box: tools/ab_env.sh VAR=VALUE [rounds] alternates tools/bench_train.py without / with it
kv=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for on in 0 1; do
    if [ $on = 1 ]; then tagname="$kv"; out=$(env $kv python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null); else tagname="(default)"; out=$(python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null); fi
    echo "$out" | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_classes']
print('$tagname'.ljust(30), round(t['ms_per_step'],3), 'ms', {a[6:]: round(v['ms_per_step'],2) for a,v in k.items()})"
  done
done
