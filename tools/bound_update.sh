#!/bin/bash
# Upper bound of fusing the DDPM / RF update into the last row kernel: the headline sampler with and without
# its update launch (T2S_SKIP_UPDATE=1: timing only, results invalid), alternated on ONE box.
cd "$(dirname "$0")/.."
rounds=${1:-2}
for r in $(seq $rounds); do
  for on in 0 1; do
    T2S_SKIP_UPDATE=$on python bench.py --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs 2>/dev/null |
      python -c "
import json,sys; t=json.loads(sys.stdin.read()); print('T2S_SKIP_UPDATE=$on', round(t['value'],3), 'series/s', round(t['ms_per_step'],2), 'ms per batch')"
  done
done
