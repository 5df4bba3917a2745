#!/bin/bash
# One-step graph replayed `steps` times (0) against the whole loop as ONE graph per lane (1): BASELINE configs[2]
# (rectified flow, B = 1024, 100 steps, cfg 5), the 32-series strong-scaling shard, and the headline (DDPM 1000, B = 256).
set -e
common="--no-cpu-baseline --no-train --no-strong --no-alt-math --no-legs"
for g in 0 1 0 1; do
  echo "== loop_graph=$g  RF B=1024 100 steps"
  python bench.py --backbone flowmatching --batch 1024 --diffusion-steps 100 --cfg-scale 5.0 --steps 6 --warmup 2 --loop-graph $g $common 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'series/s', round(d['ms_per_step'],2), 'ms/batch')"
  echo "== loop_graph=$g  RF B=32 100 steps"
  python bench.py --backbone flowmatching --batch 32 --diffusion-steps 100 --cfg-scale 5.0 --steps 20 --warmup 3 --loop-graph $g $common 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'series/s', round(d['ms_per_step'],2), 'ms/batch')"
done
for g in 0 1; do
  echo "== loop_graph=$g  DDPM B=256 1000 steps"
  python bench.py --steps 2 --warmup 1 --loop-graph $g $common 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), 'series/s', round(d['ms_per_step'],2), 'ms/batch')"
done
