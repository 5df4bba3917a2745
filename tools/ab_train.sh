#!/bin/bash
# A/B of two library builds on the same box (boxes differ by 2-3 %): tools/bin/libt2s_prev.so vs the in-tree one,
# alternating runs.  Build the baseline from another commit first:
#   git worktree add /tmp/prev <commit> && make -C /tmp/prev/t2ms_amd/csrc && cp /tmp/prev/t2ms_amd/libt2s_hip.so tools/bin/libt2s_prev.so
cp t2ms_amd/libt2s_hip.so /tmp/new.so
for rep in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/bin/libt2s_prev.so t2ms_amd/libt2s_hip.so; else cp /tmp/new.so t2ms_amd/libt2s_hip.so; fi
    echo "$which: $(timeout -k 10 300 python tools/bench_train.py --batch 1152 --steps 8 --warmup 3 --dtype bf16 --cache_latents 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))')"
  done
done
cp /tmp/new.so t2ms_amd/libt2s_hip.so
