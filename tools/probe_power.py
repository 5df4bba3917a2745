"""Does kernel time depend on the DATA (power limiting)?  Runs the f32 and the bf16x3 attention kernels on
random / zero / constant inputs; use under rocprofv3 --kernel-trace --stats (one process per data kind):
    python tools/probe_power.py randn|zeros|ones"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L

kind = sys.argv[1]
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
BH = 2048
mk = {"randn": lambda: torch.randn(BH, 480, 32, device=dev), "zeros": lambda: torch.zeros(BH, 480, 32, device=dev),
      "ones": lambda: torch.ones(BH, 480, 32, device=dev)}[kind]
q, k, v = mk(), mk(), mk()
o = torch.empty_like(q)
p = [C.c_void_p(t.data_ptr()) for t in (q, k, v, o)]
for _ in range(30):
    assert lib.t2s_attn_fwd_packed(*p, BH // 4, None) == 0
torch.cuda.synchronize()
for _ in range(30):
    assert lib.t2s_attn_fwd_x3(*p, BH, None) == 0
torch.cuda.synchronize()
print(kind, "done")
