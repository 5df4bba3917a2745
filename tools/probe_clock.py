"""Diagnostic (library built with -DT2S_EXP=32): shader clock held during the persistent attention kernel."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
n_seq = 512
q, k, v = (torch.randn(n_seq * 4, 480, 32, device=dev) for _ in range(3))
o = torch.empty(n_seq * 480 * 128, device=dev)
args = [C.c_void_p(t.data_ptr()) for t in (q, k, v, o)]
for _ in range(200):   # ~0.1 s of back-to-back launches so DVFS settles
    lib.t2s_attn_fwd_packed(*args, n_seq, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.t2s_attn_fwd_packed(*args, n_seq, None)
e1.record(); e1.synchronize()
print("avg kernel %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
n = 512 * 4 * 8
buf = (C.c_ulonglong * n)()
lib.t2s_debug_read(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(512 * 4, 8).astype(np.float64)
clk = a[:, 0] / a[:, 1] * 100e6
print("per-wave: cycles %.0f, realtime ticks %.0f -> duration %.1f us, shader clock median %.3f GHz (min %.3f max %.3f)" % (
    a[:, 0].mean(), a[:, 1].mean(), a[:, 1].mean() / 100.0, np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9))
dur = a[:, 1] / 100.0
wg = dur.reshape(512, 4).mean(axis=1)
print("per-WG duration us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (wg.min(), np.percentile(wg, 10), np.percentile(wg, 50), np.percentile(wg, 90), wg.max()))
print("first 256 WGs (first dispatched): mean %.1f ; last 256: mean %.1f" % (wg[:256].mean(), wg[256:].mean()))
hist, edges = np.histogram(wg, bins=12)
for h, e in zip(hist, edges): print("   %6.1f us: %d" % (e, h))
