#!/bin/bash
# same-box comparison of the bf16x3 sampler between the in-tree library and several variants:
#   tools/x3_ab_multi.sh [-r rounds] <lib.so ...>
cd "$(dirname "$0")/.."
rounds=2
if [ "$1" = -r ]; then rounds=$2; shift 2; fi
for r in $(seq $rounds); do for lib in "" "$@"; do
  if [ -n "$lib" ]; then export T2S_LIB=$lib; else unset T2S_LIB; fi
  python bench.py --math bf16x3 --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_breakdown_us']; print('${lib:-in-tree}'.ljust(36), round(t['value'],2), 'series/s | attention', round(k['attention_x4'],1), 'rows avg', round(k['row_chain_x5'],1), 'us')"
done; done
