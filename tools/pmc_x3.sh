#!/bin/bash
# PMC passes on the bf16x3 attention kernel (t2s_attn_fwd_x3 on 2048 heads); prints per-dispatch averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/run_x3.py <<'PY'
import ctypes as C, sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
BH = 2048
q, k, v = (torch.randn(BH, 480, 32, device=dev) for _ in range(3))
o = torch.empty_like(q)
for _ in range(3):
    assert lib.t2s_attn_fwd_x3(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()), BH, None) == 0
torch.cuda.synchronize()
PY
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS"; do
  rm -rf /tmp/pmcx
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcx -- python3 /tmp/run_x3.py > /dev/null 2> /tmp/pmcx.err || { echo "pass failed: $grp"; tail -3 /tmp/pmcx.err; continue; }
  python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('/tmp/pmcx/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'attn_fwd_x3' in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in agg.items():
    print("  %-28s %16.0f per dispatch (%d)" % (k, v / n, n))
PY
done
