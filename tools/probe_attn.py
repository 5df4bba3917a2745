"""Diagnostic: per-segment cycle shares of the attention loop (library built with -DT2S_EXP=32)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
n_seq = 512
q, k, v = (torch.randn(n_seq * 4, 480, 32, device=dev) for _ in range(3))
o = torch.empty(n_seq * 480 * 128, device=dev)
for _ in range(3):
    lib.t2s_attn_fwd_packed(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()), n_seq, None)
torch.cuda.synchronize()
n = 2 * 4096 * 4 * 8
buf = (C.c_ulonglong * n)()
lib.t2s_debug_read(buf, n)
allb = np.frombuffer(buf, dtype=np.uint64).reshape(2 * 4096 * 4, 8).astype(np.float64)
a = allb[:4096 * 4]
b = allb[4096 * 4:]
names = ["top: V read + QK(32 mfma)", "exp_sum x2 + check", "PV_A (16 mfma)", "wait vmcnt", "barrier", "issue+Kread+PV_B(16 mfma)"]
tot = a[:, 6]
print("waves:", len(a), " mean wave loop cycles: %.0f  (15 blocks)  => %.0f per block; pure MFMA would be 4096" % (tot.mean(), tot.mean() / 15))
for i, nme in enumerate(names):
    print("  %-32s mean %8.0f cyc/block  share %5.1f%%" % (nme, a[:, i].mean() / 15, 100 * a[:, i].sum() / tot.sum()))
st = a[:, 7]
print("start-time spread (cycles): min %.0f max %.0f" % (st.min() - st.min(), st.max() - st.min()))

entry, exit_ = b[:, 0], b[:, 1]
loop_start = a[:, 7]
loop_end = a[:, 7] + a[:, 6]
print("per wave: entry->loop %.0f cyc, loop %.0f, loop_end->exit(stores drained) %.0f, total %.0f" % (
    (loop_start - entry).mean(), a[:, 6].mean(), (exit_ - loop_end).mean(), (exit_ - entry).mean()))
T0 = entry.min()
print("kernel span: %.0f cycles (last exit - first entry)" % (exit_.max() - T0))
# timeline: number of waves alive in the loop vs alive at all, sampled
ts = np.linspace(T0, exit_.max(), 21)
for t in ts:
    alive = ((entry <= t) & (exit_ > t)).sum()
    inloop = ((loop_start <= t) & (loop_end > t)).sum()
    print("  t=%9.0f alive=%5d inloop=%5d" % (t - T0, alive, inloop))
