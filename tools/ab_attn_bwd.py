#!/usr/bin/env python3
"""The one-pass attention backward (T2S_ATTN_BWD_FUSED=1, csrc/t2s_attn_bf16.hip) against the two-kernel backward on the
same bf16 training step: per-tensor relative difference of all 48 gradients, and bitwise reproducibility of the fused run.
    python tools/ab_attn_bwd.py [--batch 16]"""
import argparse
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from t2ms_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    from model.denoiser.transformer import Transformer
    from t2ms_amd.train import _trainable, mse_loss
    dev = torch.device("cuda", 0)
    B = args.batch
    m = Transformer()
    m.load_state_dict(synth.make_dit_state_dict(2025), strict=True)
    m = m.to(dev).train().set_train_dtype("bf16")
    x, text = synth.make_latents(5, B).to(dev), synth.make_text_embeddings(5, B).to(dev)
    t = (torch.arange(B) % 100).to(dev)
    target = synth.make_latents(6, B).to(dev)

    def grads(fused):
        os.environ["T2S_ATTN_BWD_FUSED"] = "1" if fused else "0"
        m.zero_grad()
        loss = mse_loss(m(input=x, t=t, text_input=text), target)
        loss.backward()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in _trainable(m)]

    two = grads(False)
    one = grads(True)
    again = grads(True)
    worst = 0.0
    for i, (a, b, c) in enumerate(zip(two, one, again)):
        rel = float((a - b).norm() / (a.norm() + 1e-30))
        worst = max(worst, rel)
        assert torch.isfinite(b).all(), i
        assert torch.equal(b, c), f"tensor {i}: fused backward not reproducible"
    print(f"B={B}: worst relative difference fused vs two kernels over {len(two)} gradients: {worst:.3e}; fused run reproducible")
    assert worst < 5e-3


if __name__ == "__main__":
    main()
