"""The CPU oracle vs the golden vectors produced by running the reference
(tests/golden/gen_golden.py).  No GPU needed."""
import os

import numpy as np
import pytest
import torch

from oracle import t2s_oracle as O
from t2ms_amd import synth


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name + ".npz")).items()}


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _maxdiff(a, b):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().numpy() if torch.is_tensor(b) else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def _close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_ddpm_tables_and_samples(golden_dir):
    g = _load(golden_dir, "ddpm")
    for T in (50, 1000):
        tab = O.ddpm_tables(T)
        for k in ("beta", "alpha", "alpha_bar"):
            assert np.array_equal(tab[k].numpy(), g[f"{k}_{T}"]), (k, T)
    tab = O.ddpm_tables(1000)
    t = _t(g["t"])
    assert np.array_equal(O.ddpm_q_sample(tab, _t(g["x0"]), t, _t(g["eps"])).numpy(), g["q_sample"])
    assert np.array_equal(
        O.ddpm_p_sample(tab, _t(g["x0"]), _t(g["eps_hat"]), t, _t(g["p_noise"])).numpy(), g["p_sample"])


def test_rectified_flow(golden_dir):
    g = _load(golden_dir, "rf")
    assert np.array_equal(O.rf_euler(_t(g["x1"]), _t(g["v"]), 1.0 / 100).numpy(), g["euler"])
    assert np.array_equal(O.rf_create_flow(_t(g["x1"]), _t(g["t"]), _t(g["x_0"])).numpy(), g["x_t"])


def test_time_and_pos_embedding(golden_dir):
    g = _load(golden_dir, "time_emb")
    assert np.array_equal(O.time_embedding(_t(g["t_long"])).numpy(), g["emb_long"])
    assert np.array_equal(O.time_embedding(_t(g["t_float"])).numpy(), g["emb_float"])
    assert np.array_equal(O.sinusoidal_pos_embed().numpy(), g["pos_embed"])
    assert np.array_equal(synth.make_dit_state_dict(1)["pos_embed"].numpy(), g["pos_embed"])


def test_dit_forward(golden_dir):
    g = _load(golden_dir, "dit_forward")
    sd = synth.make_dit_state_dict(2025)
    x = synth.make_latents(2025, 4)
    text = synth.make_text_embeddings(2025, 4)
    taps = {}
    with torch.no_grad():
        yc = O.dit_forward(sd, x, _t(g["t_long"]), text, taps)
        yu = O.dit_forward(sd, x, _t(g["t_long"]), None)
        yf = O.dit_forward(sd, x, _t(g["t_float"]), text)
    # fused SDPA (reference run) vs explicit softmax (oracle): rounding-level differences only
    _close(yc, g["cond"], atol=2e-5)
    _close(yu, g["uncond"], atol=2e-5)
    _close(yf, g["cond_float"], atol=2e-5)
    for i in range(4):
        _close(taps[f"post_mlp_{i}"][:1, ::7], g[f"tap_post_mlp_{i}"], atol=2e-5)
    # the fixture is not vacuous: gates are non-zero so blocks change the stream
    assert float(np.abs(g["tap_post_mlp_3"] - g["tap_post_mlp_0"]).max()) > 1e-2
    assert float(np.abs(g["cond"] - g["uncond"]).max()) > 1e-3


def test_timm_boundary_restatement(golden_dir):
    """UNPINNED boundary: both sides are restatements of timm 1.0.11 (gen script: fused SDPA
    form; oracle: explicit softmax form)."""
    g = _load(golden_dir, "timm_boundary")
    sd = synth.make_dit_state_dict(2025)
    xa = torch.from_numpy(np.random.RandomState(4242).randn(2, 480, 128).astype(np.float32))
    p = "layers.0."
    ya = O.timm_attention(xa, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                          sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    ym = O.timm_mlp(xa, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"],
                    sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    _close(ya[:, ::5], g["attn_out"], atol=1e-5)
    _close(ym[:, ::5], g["mlp_out"], atol=1e-5)


def test_timm_restatement_equals_torch_library_modules():
    """timm is absent, so its two leaves cannot be run here -- but torch's own library modules implement the
    same published mathematics independently of this repo: `nn.MultiheadAttention(128, 4, batch_first=True)` with
    in_proj = qkv (rows [q | k | v], heads = contiguous 32-feature slices, scale 32^-0.5) and out_proj = proj is timm's
    Attention(128, num_heads=4, qkv_bias=True); Linear -> GELU(tanh) -> Linear is timm's Mlp.  The oracle's restatements
    (both attention branches) against those modules: a third-party cross-check, still not a pin on timm itself."""
    sd = synth.make_dit_state_dict(2025)
    p = "layers.1."
    x = torch.from_numpy(np.random.RandomState(77).randn(3, 480, 128).astype(np.float32))
    mha = torch.nn.MultiheadAttention(128, 4, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(sd[p + "attn.qkv.weight"])
        mha.in_proj_bias.copy_(sd[p + "attn.qkv.bias"])
        mha.out_proj.weight.copy_(sd[p + "attn.proj.weight"])
        mha.out_proj.bias.copy_(sd[p + "attn.proj.bias"])
        ref, _ = mha(x, x, x, need_weights=False)
        mlp = torch.nn.Sequential(torch.nn.Linear(128, 256), torch.nn.GELU(approximate="tanh"), torch.nn.Linear(256, 128))
        mlp[0].weight.copy_(sd[p + "mlp.fc1.weight"])
        mlp[0].bias.copy_(sd[p + "mlp.fc1.bias"])
        mlp[2].weight.copy_(sd[p + "mlp.fc2.weight"])
        mlp[2].bias.copy_(sd[p + "mlp.fc2.bias"])
        ref_mlp = mlp(x)
        args = (x, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        for impl in ("explicit", "sdpa"):
            O.set_attention_impl(impl)
            try:
                _close(O.timm_attention(*args), ref.numpy(), atol=2e-6)
            finally:
                O.set_attention_impl("explicit")
        _close(O.timm_mlp(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "mlp.fc2.weight"],
                          sd[p + "mlp.fc2.bias"]), ref_mlp.numpy(), atol=2e-6)


@pytest.mark.parametrize("L", [24, 48, 96])
@pytest.mark.parametrize("B", [1, 5])
def test_vae(golden_dir, L, B):
    g = _load(golden_dir, "vae")
    vsd = synth.make_vae_state_dict(2025)
    xs = synth.make_series(100 + L + B, B, L)
    with torch.no_grad():
        z, before = O.vae_encode(vsd, xs)
        rec, after = O.vae_decode(vsd, z, L)
        rec2, _ = O.vae_decode(vsd, synth.make_latents(300 + L, B), L)
    _close(z, g[f"z_{L}_{B}"], atol=1e-6)
    _close(before, g[f"before_{L}_{B}"], atol=1e-6)
    _close(after, g[f"after_{L}_{B}"], atol=1e-6)
    assert tuple(rec.shape) == g[f"rec_{L}_{B}"].shape == ((L,) if B == 1 else (B, L))
    _close(rec, g[f"rec_{L}_{B}"], atol=1e-6)
    _close(rec2, g[f"rec_rand_{L}_{B}"], atol=1e-5)


@pytest.mark.parametrize("L", [512, 2048])
@pytest.mark.parametrize("B", [1, 3])
def test_vae_long_series(golden_dir, L, B):
    """vqvae.py accepts any L (the reference's SUSHI set is 2048 long): the oracle against the reference run at L = 512 and
    2048 (tests/golden/gen_golden_r3.py) -- what the GPU's time-tiled LA-VAE kernels are checked against."""
    g = _load(golden_dir, "vae_long")
    vsd = synth.make_vae_state_dict(2025)
    xs = synth.make_series(100 + L + B, B, L)
    st = 7 if L == 512 else 11
    with torch.no_grad():
        z, before = O.vae_encode(vsd, xs)
        rec, after = O.vae_decode(vsd, z, L)
    assert tuple(rec.shape) == g[f"rec_{L}_{B}"].shape
    assert _maxdiff(z, g[f"z_{L}_{B}"]) < 1e-6 and _maxdiff(rec, g[f"rec_{L}_{B}"]) < 1e-6
    assert _maxdiff(before[:, :, ::st], g[f"before_s{st}_{L}_{B}"]) < 1e-6
    assert _maxdiff(after[:, :, ::st], g[f"after_s{st}_{L}_{B}"]) < 1e-6
    assert np.allclose(before.double().sum(2).numpy(), g[f"before_rowsum_{L}_{B}"], rtol=0, atol=1e-9)



def test_chains(golden_dir):
    g = _load(golden_dir, "chains")
    sd = synth.make_dit_state_dict(31337, gain=0.7)
    vsd = synth.make_vae_state_dict(2025)
    xT = synth.make_latents(31337, 4)
    text = synth.make_text_embeddings(31337, 4)
    noises = torch.from_numpy(np.random.RandomState(99).randn(20, 4, 64, 30).astype(np.float32))
    with torch.no_grad():
        xd = O.sample_ddpm(sd, xT, text, 20, 7.0, noises)
        xr = O.sample_rf(sd, xT, text, 20, 7.0)
        sd_, _ = O.vae_decode(vsd, xd, 96)
        sr_, _ = O.vae_decode(vsd, xr, 96)
    scale = float(np.abs(g["ddpm_latent"]).max())
    _close(xd, g["ddpm_latent"], atol=1e-4 * max(1.0, scale))
    _close(xr, g["rf_latent"], atol=1e-4 * max(1.0, float(np.abs(g["rf_latent"]).max())))
    _close(sd_, g["ddpm_series"], atol=1e-4 * max(1.0, scale))
    _close(sr_, g["rf_series"], atol=1e-4 * max(1.0, scale))


def test_mlp_denoiser(golden_dir):
    g = _load(golden_dir, "mlp_denoiser")
    msd = synth.make_mlp_state_dict(2025)
    t = torch.tensor([49, 20, 1, 0])
    with torch.no_grad():
        yc = O.mlp_denoiser_forward(msd, _t(g["x"]), t, synth.make_text_embeddings(5, 4))
        yu = O.mlp_denoiser_forward(msd, _t(g["x"]), t, None)
    _close(yc, g["cond"], atol=1e-5, rtol=1e-5)
    _close(yu, g["uncond"], atol=1e-5, rtol=1e-5)


def test_train_step_grads(golden_dir):
    """Autograd through the oracle reproduces the reference's loss and per-parameter grad norms
    (fixture 9; anchors the later backward kernels)."""
    g = _load(golden_dir, "train_step")
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in synth.make_dit_state_dict(2025).items()}
    tab = O.ddpm_tables(100)
    x1 = synth.make_latents(555, 4)
    tt = torch.tensor([3, 50, 77, 99])
    eps = synth.make_latents(556, 4)
    xt = O.ddpm_q_sample(tab, x1, tt, eps)
    pred = O.dit_forward(sd, xt, tt, synth.make_text_embeddings(555, 4))
    loss = O.mse_loss(pred, eps)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    n = 0
    for k, v in sd.items():
        key = "gn_" + k.replace(".", "__")
        if key in g:
            np.testing.assert_allclose(v.grad.norm().item(), g[key], rtol=2e-4, atol=1e-7)
            n += 1
        else:
            assert v.grad is None or k.startswith("unpatch") or float(v.grad.abs().max()) == 0.0
    assert n == 48
    _close(sd["layers.0.attn.qkv.weight"].grad[::16, ::8], g["grad_qkv0_sample"], atol=1e-6, rtol=1e-3)


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert tuple(int(x) for x in got) == want
    z = O.device_normal(2025, 3, 0, 64)
    assert z.shape == (64, 1920) and abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02


def test_eval_metrics_restatement():
    """oracle.eval_mse / eval_wape (evaluation.py:166-206) on a hand-checkable case incl. the all-zero row."""
    import numpy as np
    ori = np.zeros((3, 4, 1)); gen = np.zeros((3, 4, 1))
    ori[0, :, 0] = [1, 2, 3, 4]; gen[0, :, 0] = [1, 2, 3, 6]        # mse 1.0, wape 2/10
    ori[1, :, 0] = [0, 0, 0, 0]; gen[1, :, 0] = [1, 1, 1, 1]        # mse 1.0, wape NaN (skipped)
    ori[2, :, 0] = [2, 2, 2, 2]; gen[2, :, 0] = [1, 1, 1, 1]        # mse 1.0, wape 0.5
    assert abs(O.eval_mse(ori, gen) - 1.0) < 1e-12
    assert abs(O.eval_wape(ori, gen) - 0.35) < 1e-12


def test_eval_mrr_restatement():
    """oracle.eval_mrr (evaluation.py:21-45) on hand-checkable cases: the score is 1 / (run index + 1) of the best
    run when it clears the threshold, ties go to the later run, a zero vector has similarity 0."""
    import numpy as np
    ori = np.zeros((4, 3, 1)); gen = np.zeros((4, 3, 1, 3))
    ori[0, :, 0] = [1, 0, 0]; gen[0, :, 0, :] = np.array([[0, 1, 0], [1, 1, 0], [1, 0, 0]]).T   # best = run 2 -> 1/3
    ori[1, :, 0] = [1, 0, 0]; gen[1, :, 0, :] = np.array([[2, 0, 0], [0, 1, 0], [3, 0, 0]]).T   # tie 0 / 2 -> run 2 -> 1/3
    ori[2, :, 0] = [1, 0, 0]; gen[2, :, 0, :] = np.array([[1, 2, 0], [0, 1, 0], [0, 0, 1]]).T   # best 0.447 < 0.5 -> 0
    ori[3, :, 0] = [0, 0, 0]; gen[3, :, 0, :] = 1.0                                              # zero vector -> 0
    assert abs(O.eval_cosine([1, 0, 0], [1, 1, 0]) - 2 ** -0.5) < 1e-12
    assert O.eval_cosine([0, 0], [1, 1]) == 0.0
    assert abs(O.eval_mrr(ori, gen) - (1 / 3 + 1 / 3) / 4) < 1e-12
    gen[1, :, 0, 0] = [2, 0, 0]; gen[1, :, 0, 2] = [3, 0.1, 0]                                   # run 0 now best -> 1
    assert abs(O.eval_mrr(ori, gen) - (1 / 3 + 1.0) / 4) < 1e-12
