#!/bin/bash
# experiment: bgemm grid size (persistent vs one tile per wave)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for g in "$@"; do
  export T2S_BG_GRID=$g
  rm -rf /tmp/tp_$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_$g -- python3 tools/bench_train.py --batch 1152 --steps 2 --warmup 1 --dtype bf16 --cache_latents > gpurun_out/bg_$g.log 2>&1 || exit 1
  echo "== grid $g: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bg_$g.log)"
  grep -E "bgemm|wgrad16" /tmp/tp_$g/*/*kernel_stats.csv | awk -F, '{printf "%s %s %.1f us\n", $1, $2, $4/1000}' | sed 's/.*kernel_stats.csv://'
done
