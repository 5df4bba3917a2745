#!/bin/bash
# PMC passes on the row-chain kernels inside a short bench run (3 diffusion steps); per-dispatch averages per kernel
# usage: tools/pmc_rows.sh [f32|bf16x3]
math=${1:-f32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pmcr
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcr -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --diffusion-steps 3 --no-cpu-baseline --no-alt-math --math $math > /dev/null 2> /tmp/pmcr.err || { echo "pass failed: $grp"; tail -3 /tmp/pmcr.err; continue; }
  python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('/tmp/pmcr/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'dit_rows' in r['Kernel_Name']:
        a = agg[(r['Kernel_Name'][:48], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("  %-50s %-28s %14.0f (%d)" % (k, c, v / n, n))
PY
done
