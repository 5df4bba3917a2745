#!/bin/bash
# SQ counter passes over the bf16 training step (tools/bench_train.py); per-dispatch averages for kernels matching $1
pat=${1:-attn16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
  rm -rf /tmp/pmct
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmct -- python3 tools/bench_train.py --steps 1 --warmup 1 > /dev/null 2> /tmp/pmct.err || { echo "pass failed: $grp"; tail -3 /tmp/pmct.err; continue; }
  PAT=$pat python3 - <<'PY'
import csv, glob, collections, os
f = sorted(glob.glob('/tmp/pmct/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if os.environ['PAT'] in r['Kernel_Name']:
        a = agg[(r['Kernel_Name'][:36], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("  %-38s %-28s %16.0f (%d)" % (k, c, v / n, n))
PY
done
