#!/bin/bash
# diagnostic: rebuild the library with -DT2S_EXP=n variants on the GPU box and time the row kernels
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  make -C t2ms_amd/csrc clean >/dev/null; make -C t2ms_amd/csrc -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DT2S_EXP=$e" >/dev/null 2>&1
  rm -rf /tmp/p$e; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$e -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --diffusion-steps 10 --no-cpu-baseline >/dev/null 2>&1)
  echo "EXP=$e"; python3 - <<PY
import csv,glob
f=sorted(glob.glob('/tmp/p$e/*/*kernel_stats.csv'))[-1]
for r in [r for r in csv.DictReader(open(f)) if "rows" in r["Name"] or "attn" in r["Name"]]:
    print("   %-60s %8.1f us" % (r['Name'][:60], float(r['AverageNs'])/1e3))
PY
done
