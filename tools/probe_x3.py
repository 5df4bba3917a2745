"""Diagnostic: per-segment cycle shares of the bf16x3 attention loop (library built with -DT2S_EXP=512)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
BH = 2048
q, k, v = (torch.randn(BH, 480, 32, device=dev) for _ in range(3))
o = torch.empty_like(q)
for _ in range(2):
    rc = lib.t2s_attn_fwd_x3(C.c_void_p(q.data_ptr()), C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(o.data_ptr()), BH, None)
    assert rc == 0
torch.cuda.synchronize()
n = 8 * 8 * 256
buf = (C.c_ulonglong * n)()
assert lib.t2s_debug_read_x3(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
names = ["QK (24 mfma)", "DMA wait (vmcnt) before the barrier", "V loads + exp+sum+check", "split P", "s_barrier + DMA issue", "Kload+PV (24 mfma)+prefetch", "head end (O store, Q split)"]
for grp, ws in (("waves 0-3 (STAG=0)", slice(0, 4)), ("waves 4-6 (STAG=1)", slice(4, 7)), ("wave 7 (NT=1)", slice(7, 8))):
    g = a[:, ws, :].reshape(-1, 8)
    tot = g[:, 7].mean()
    heads = 8.0
    print(f"{grp}: total {tot:.0f} cycles for {heads:.0f} heads = {tot / heads / 15:.0f} per key block")
    for i, nme in enumerate(names):
        per = g[:, i].mean() / heads / (15 if i < 6 else 1)
        print(f"   {nme:32s} {per:8.0f} cyc/{'block' if i < 6 else 'head'}  {100 * g[:, i].mean() / tot:5.1f}%")
