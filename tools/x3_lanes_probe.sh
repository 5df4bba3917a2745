#!/bin/bash
# series/s of the bf16x3 sampler by batch and lane count (the automatic lane rule of t2s_sampler_run was tuned on the f32 kernels)
cd "$(dirname "$0")/.."
for b in ${1:-32 64 96 128 192 256}; do for l in 1 2 3; do
python bench.py --math bf16x3 --batch $b --lanes $l --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); print('bf16x3 batch $b lanes $l', round(t['value'],2), 'series/s')"
done; done
