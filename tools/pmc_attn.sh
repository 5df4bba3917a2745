#!/bin/bash
# PMC passes on the f32 attention kernel (t2s_attn_fwd_packed, 512 sequences = 2048 heads)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/run_attn.py <<'PY'
import ctypes as C, sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
n_seq = 512
q, k, v = (torch.randn(n_seq * 4, 480, 32, device=dev) for _ in range(3))
o = torch.empty(n_seq * 480 * 128, device=dev)
for _ in range(3):
    assert lib.t2s_attn_fwd_packed(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()), n_seq, None) == 0
torch.cuda.synchronize()
PY
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA"; do
  rm -rf /tmp/pmca
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmca -- python3 /tmp/run_attn.py > /dev/null 2> /tmp/pmca.err || { echo "pass failed: $grp"; tail -3 /tmp/pmca.err; continue; }
  python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('/tmp/pmca/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if 'attn_fwd_persistent' in r['Kernel_Name']:
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in agg.items():
    print("  %-28s %16.0f per dispatch (%d)" % (k, v / n, n))
PY
done
