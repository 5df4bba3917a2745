"""Diagnostic (library built with -DT2S_EXP=64): per-phase cycle shares of dit_rows_kernel<true,true>."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L, synth
from model.denoiser.transformer import Transformer
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
m = Transformer(); m.load_state_dict(synth.make_dit_state_dict(2025), strict=True); m = m.to(dev).eval()
if len(sys.argv) > 1: m.set_math(sys.argv[1])   # "bf16x3": probe dit_rows_x3_kernel<true,true> instead
B = 256
x = synth.make_latents(1, B).to(dev); text = synth.make_text_embeddings(1, B).to(dev)
h = m.t2s_handle(dev, 2 * B)
temb = m.time_emb(torch.full((1,), 500, device=dev))
ou, oc = torch.empty_like(x), torch.empty_like(x)
for _ in range(3):
    L.check(L.lib().t2s_dit_forward_cfg(h, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(), oc.data_ptr(), B, None))
torch.cuda.synchronize()
n = 1920 * 4 * 8
buf = (C.c_ulonglong * n)()
lib.t2s_debug_read_rows(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(1920 * 4, 8).astype(np.float64)
names = ["prologue: DMA0 + consts + x/ao loads + barrier", "proj: 4 chunks (256 MFMA = 16384 cyc)", "LN2 + modulate + park x",
         "(stamp)", "MLP: 16 chunks (1024 MFMA = 65536 cyc) + residual", "LN1' + modulate", "QKV: 12 chunks (768 MFMA = 49152 cyc)", "drain stores"]
d = np.diff(a, axis=1)
tot = a[:, 7] - a[:, 0]
print("waves %d  mean lifetime %.0f cycles (pure MFMA 131072 per wave; 2 waves/SIMD => >= 262144 when co-resident)" % (len(a), tot.mean()))
for i in range(7):
    print("  %-58s mean %9.0f  share %5.1f%%" % (names[i] if i < 3 else names[i + 1] if i >= 3 else "", d[:, i].mean(), 100 * d[:, i].sum() / tot.sum()))
print("lifetime percentiles:", [int(np.percentile(tot, p)) for p in (1, 10, 50, 90, 99)])
