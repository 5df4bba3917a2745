"""BASELINE configs[0] on the GPU: the MLP denoiser's forward as one HIP launch (t2s_mlp_forward) against the same mirror's
torch-op layers, and the 50-step DDPM loop of infer.py --denoiser MLP (B = 32, L = 24) either way.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from t2ms_amd import synth                      # noqa: E402
from model.denoiser.mlp import MLP              # noqa: E402
from model.backbone.DDPM import DDPM            # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    m = MLP().eval()
    m.load_state_dict(synth.make_mlp_state_dict(2025), strict=True)
    m = m.to(dev)
    out = {}
    for B in (32, 64, 256, 1024):
        x = torch.randn(B, 64, 6, device=dev)
        t = torch.full((B,), 25, device=dev)
        text = synth.make_text_embeddings(1, B).to(dev)

        def torch_layers():
            h = x
            for layer in m.layers:
                h = layer(h, t, text)
            return h
        from t2ms_amd import _lib as L
        from t2ms_amd.model.denoiser.mlp import _freqs_on
        with torch.no_grad():
            m(x, t, text)
        packed, fr, tf, y = m.__dict__["_t2s_packed"][2], _freqs_on(dev), t.float(), torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        out[f"kernel_us_B{B}"] = round(1e6 * timed(lambda: L.lib().t2s_mlp_forward(
            packed.data_ptr(), x.data_ptr(), tf.data_ptr(), fr.data_ptr(), text.data_ptr(), y.data_ptr(), B, st), 500), 1)
        with torch.no_grad():
            out[f"forward_us_hip_B{B}"] = round(1e6 * timed(lambda: m(x, t, text), 200), 1)
            out[f"forward_us_torch_ops_B{B}"] = round(1e6 * timed(torch_layers, 20), 1)
    B, steps, cfg = 32, 50, 7.0
    ddpm = DDPM(steps, dev)
    text = synth.make_text_embeddings(2, B).to(dev)

    def loop(forward):
        x = torch.randn(B, 64, 6, device=dev)
        for j in range(steps):
            t = torch.full((B,), steps - 1 - j, dtype=torch.long, device=dev)
            u, c = forward(x, t, None), forward(x, t, text)
            x = ddpm.p_sample(x, u + cfg * (c - u), t)
        return x

    def torch_forward(x, t, tx):
        for layer in m.layers:
            x = layer(x, t, tx)
        return x
    with torch.no_grad():
        out["loop50_ms_hip"] = round(1e3 * timed(lambda: loop(m), 5), 2)
        out["loop50_ms_torch_ops"] = round(1e3 * timed(lambda: loop(torch_forward), 2), 2)
    # one training forward + backward at B = 32 (train.py --denoiser MLP): HIP autograd node against torch-op autograd
    B = 32
    xg = torch.randn(B, 64, 6, device=dev)
    tt = torch.randint(0, 100, (B,), device=dev)
    wgt = torch.randn(B, 64, 6, device=dev)

    def train_pass():
        for p in m.parameters():
            p.grad = None
        (m(xg, tt, text) * wgt).sum().backward()
    m.train()
    out["fwd_bwd_ms_hip_B32"] = round(1e3 * timed(train_pass, 20), 3)
    os.environ["T2S_MLP_TORCH_AUTOGRAD"] = "1"
    out["fwd_bwd_ms_torch_ops_B32"] = round(1e3 * timed(train_pass, 5), 3)
    os.environ.pop("T2S_MLP_TORCH_AUTOGRAD")
    m.eval()
    out["series_per_s_hip"] = round(B / (out["loop50_ms_hip"] / 1e3), 1)
    out["series_per_s_torch_ops"] = round(B / (out["loop50_ms_torch_ops"] / 1e3), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
