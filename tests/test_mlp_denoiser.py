"""a20 / BASELINE config 1 (plumbing): the MLP-denoiser mirror vs the golden vectors produced by the
reference, and the runnable 50-step DDPM equivalent SURVEY.md 8(d) prescribes (latent = the
pre-interpolation `before` (B,64,6) of an L=24 encode)."""
import os

import numpy as np
import torch

from oracle import t2s_oracle as O
from t2ms_amd import synth


def test_mlp_mirror_matches_reference_golden(golden_dir):
    from model.denoiser.mlp import MLP
    g = np.load(os.path.join(golden_dir, "mlp_denoiser.npz"))
    m = MLP().eval()
    assert set(m.state_dict()) == set(synth.make_mlp_state_dict(1))
    m.load_state_dict(synth.make_mlp_state_dict(2025), strict=True)
    assert sum(p.numel() for p in m.parameters()) == 632784
    t = torch.tensor([49, 20, 1, 0])
    with torch.no_grad():
        yc = m(torch.from_numpy(g["x"]), t, synth.make_text_embeddings(5, 4))
        yu = m(torch.from_numpy(g["x"]), t, None)
    np.testing.assert_allclose(yc.numpy(), g["cond"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(yu.numpy(), g["uncond"], atol=1e-5, rtol=1e-5)


def test_config1_plumbing_chain_matches_oracle():
    """ETTh1-like L=24, MLP denoiser, 50-step DDPM with CFG, B=32 on the CPU (config 1)."""
    from model.denoiser.mlp import MLP
    msd = synth.make_mlp_state_dict(2025)
    m = MLP().eval()
    m.load_state_dict(msd, strict=True)
    B, steps, cfg = 32, 50, 7.0
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.randn(B, 64, 6).astype(np.float32))
    x_ref = x.clone()
    text = synth.make_text_embeddings(2025, B)
    noises = torch.from_numpy(rs.randn(steps, B, 64, 6).astype(np.float32))
    tab = O.ddpm_tables(steps)
    with torch.no_grad():
        for j in range(steps):
            t = torch.full((B,), steps - 1 - j, dtype=torch.long)
            pred = m(x, t, None)
            pred = pred + cfg * (m(x, t, text) - pred)
            x = O.ddpm_p_sample(tab, x, pred, t, noises[j])
            u = O.mlp_denoiser_forward(msd, x_ref, t, None)
            c = O.mlp_denoiser_forward(msd, x_ref, t, text)
            x_ref = O.ddpm_p_sample(tab, x_ref, u + cfg * (c - u), t, noises[j])
    scale = max(1.0, float(x_ref.abs().max()))
    assert float((x - x_ref).abs().max()) < 1e-4 * scale and bool(torch.isfinite(x).all())
