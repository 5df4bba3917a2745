"""Experiment: does running the CFG pass as two independent half batches on two streams (two handles) beat one
full-batch pass?  Kernel boundaries drain and refill the chip and the row-chain kernel has a 7.5-tiles-per-SIMD
quantisation (1920 workgroups over 512 slots); a second stream can fill those holes.
    python tools/exp_twostream.py [f32|bf16x3] [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import _lib as L, synth
from model.denoiser.transformer import Transformer

math = sys.argv[1] if len(sys.argv) > 1 else "f32"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
B = 256


def make():
    m = Transformer()
    m.load_state_dict(synth.make_dit_state_dict(2025), strict=True)
    m = m.to(dev).eval()
    if math != "f32":
        m.set_math(math)
    return m


x = synth.make_latents(1, B).to(dev)
text = synth.make_text_embeddings(1, B).to(dev)
full = make()
hf = full.t2s_handle(dev, 2 * B)
temb = full.time_emb(torch.full((1,), 500, device=dev))
ou, oc = torch.empty_like(x), torch.empty_like(x)
lib = L.lib()


def run_full(n):
    for _ in range(n):
        L.check(lib.t2s_dit_forward_cfg(hf, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(), oc.data_ptr(), B, None))


def timed(fn, n):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_full = timed(run_full, iters)
print(f"{math}: one stream, B={B}: {t_full:.3f} ms per CFG pass")

for sizes in ((128, 128), (160, 96), (192, 64), (96, 96, 64), (128, 64, 64), (64, 64, 64, 64), (96, 64, 64, 32)):
    parts = len(sizes)
    starts = [sum(sizes[:p]) for p in range(parts)]
    models = [make() for _ in range(parts)]
    handles = [m.t2s_handle(dev, 2 * sizes[p]) for p, m in enumerate(models)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    ou2, oc2 = torch.empty_like(x), torch.empty_like(x)
    esz = x[0].numel() * 4

    def run_parts(n):
        for _ in range(n):
            for p in range(parts):
                off = starts[p] * esz
                L.check(lib.t2s_dit_forward_cfg(handles[p], x.data_ptr() + off, temb.data_ptr(), text.data_ptr() + starts[p] * 128 * 4,
                                                ou2.data_ptr() + off, oc2.data_ptr() + off, sizes[p], streams[p].cuda_stream))

    torch.cuda.synchronize()
    t = timed(run_parts, iters)
    torch.cuda.synchronize()
    same = bool((ou2 == ou).all() and (oc2 == oc).all())
    print(f"{math}: streams {sizes}: {t:.3f} ms per CFG pass ({t_full / t:.3f}x), bitwise equal to one stream: {same}")
