"""Inputs and the acceptance check of the 1000-step DDPM chain fixture (tests/golden/chain1000.npz), shared by the
CPU oracle test and the GPU tests."""
import os

import numpy as np
import torch

from t2ms_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def chain1000_inputs():
    """x_T, text and the 1000 injected draws of tests/golden/chain1000.npz (the generator re-seeds torch before every
    p_sample, DDPM.py:35 draws torch.randn(xt.shape))."""
    xT = synth.make_latents(1000, 2)
    text = synth.make_text_embeddings(1000, 2)
    noises = torch.empty(1000, 2, 64, 30)
    state = torch.get_rng_state()
    for j in range(1000):
        torch.manual_seed(50_000 + j)
        noises[j] = torch.randn(2, 64, 30)
    torch.set_rng_state(state)
    return xT, text, noises


CHAIN_TAPS = (0, 1, 9, 99, 499, 998, 999)


def check_chain1000(final, series, taps, label=""):
    """final (2,64,30), series (2,96), taps {j: x after loop index j} against the reference run: 1e-4 relative to the
    state's size at that step (the untrained model does not cancel the schedule's 1/sqrt(alpha) growth: |x| reaches
    1.2e3 at the end -- SURVEY.md section 7, hard part 1).  Returns (and prints: `pytest -s`) the achieved margins as
    fractions of the tolerance; the fp64-referenced figures are in profiles/r03_accuracy.json (tools/accuracy_table.py)."""
    g = _load("chain1000")
    margins = {}
    for j, x in taps.items():
        ref = g[f"x_after_{j}"]
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        e = float(np.abs(np.asarray(x) - ref).max())
        margins[f"x_after_{j}"] = e / tol
        assert e <= tol, f"step {j}: {e:.3e} > {tol:.3e}"
    scale = float(g["max_abs"])
    assert scale > 100
    e = float(np.abs(np.asarray(final) - g["latent"]).max())
    margins["latent"] = e / (1e-4 * scale)
    assert e <= 1e-4 * scale, f"final latent: {e:.3e} > {1e-4 * scale:.3e}"
    if series is not None:
        tol = 1e-4 * max(1.0, float(np.abs(g["series"]).max()))
        e = float(np.abs(np.asarray(series) - g["series"]).max())
        margins["series"] = e / tol
        assert e <= tol, f"series: {e:.3e} > {tol:.3e}"
    print(f"chain1000 {label}: error / tolerance = " + ", ".join(f"{k} {v:.3f}" for k, v in margins.items()))
    return margins
