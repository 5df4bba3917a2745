#!/bin/bash
# Same-box A/B of two library builds on the SAMPLING path (box-to-box spread is 1.5-5 %, larger than most kernel changes):
#   tools/ab_sample.sh <other .so> [rounds] [batches]   alternates tools/strong_probe.py between the in-tree build and T2S_LIB=<other>
other=${1:?other .so}; rounds=${2:-2}; batches=${3:-256,128,32}
for r in $(seq $rounds); do
  for lib in "" "$other"; do
    if [ -n "$lib" ]; then export T2S_LIB=$lib; else unset T2S_LIB; fi
    python tools/strong_probe.py --batches $batches --reps 2 2>/dev/null | grep "^[0-9]" | python -c "
import json,sys
for ln in sys.stdin:
    b, js = ln.split(' ', 1); d = json.loads(js)
    print('${lib:-in-tree}'.ljust(30), 'B=%-4s %.2f series/s  step %.1f us  attn %.1f rows %.1f other %.1f' % (b, d['series_per_s'], d['ms_per_cfg_step']*1e3, d['attn_us'], d['rows_us'], d['other_us']))"
  done
done
