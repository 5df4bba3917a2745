"""a20 / BASELINE config 1 (plumbing): the MLP-denoiser mirror vs the golden vectors produced by the
reference, and the runnable 50-step DDPM equivalent SURVEY.md 8(d) prescribes (latent = the
pre-interpolation `before` (B,64,6) of an L=24 encode)."""
import os

import numpy as np
import pytest
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


def test_latent_cache_plumbing_cpu():
    """datasets with attached latents hand out row indices; the mixed collate keeps them
    per length group; encode_all chunks through the encoder (a stand-in here: no GPU on this tier)."""
    import numpy as np
    from datafactory.dataset import SyntheticT2SDataset
    from datafactory.dataloader import AlternatingDataset, custom_collate_fn
    from t2ms_amd import latent_cache
    parts = [SyntheticT2SDataset(10, L, seed=L) for L in (24, 48, 96)]
    ds = AlternatingDataset(*parts)
    assert len(ds[0][0]) == 3                                   # (text, x, emb) without a cache
    enc = lambda x: (x.mean(dim=1, keepdim=True).repeat(1, 4), None)    # noqa: E731
    by_len = latent_cache.attach(ds, enc, torch.device("cpu"))
    assert sorted(by_len) == [24, 48, 96] and by_len[48].shape == (10, 4)
    batch = [ds[i] for i in (0, 3, 12, 25, 29)]
    groups = custom_collate_fn(batch)
    assert [g[1].shape[1] for g in groups] == [24, 48, 96]
    for texts, xs, embs, idx in groups:
        assert idx.dtype == torch.long and len(texts) == xs.shape[0] == embs.shape[0] == idx.shape[0]
        z = by_len[int(xs.shape[1])][idx]
        np.testing.assert_allclose(z[:, 0].numpy(), xs.float().mean(dim=1).numpy(), rtol=1e-6)
    assert groups[1][3].tolist() == [2] and groups[2][3].tolist() == [5, 9]
    with pytest.raises(ValueError):
        parts[0].attach_latents(torch.zeros(3, 4))


@pytest.mark.gpu
def test_config1_end_to_end_through_infer_driver(tmp_path, monkeypatch):
    """BASELINE configs[0] wired end to end: `infer.py --denoiser MLP` on L=24 series, B=32,
    50-step DDPM with CFG -- HIP encoder -> `before` (32,64,6) -> MLP denoiser (t2s_mlp_forward) + HIP DDPM update on the 6-wide
    latent -> HIP Decoder(length=24) -- against the CPU oracle run on the same rows, draws (Philox restated in numpy)
    and weights; the four .npy files keep the reference layout."""
    import infer as drv
    from datafactory.dataset import SyntheticT2SDataset
    assert torch.cuda.is_available()
    monkeypatch.chdir(tmp_path)
    save = str(tmp_path / "results")
    steps, cfg, B, seed = 50, 7.0, 32, 3
    drv.main(["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "MLP", "--total_step", str(steps),
              "--cfg_scale", str(cfg), "--batch_size", str(B), "--save_path", save, "--synthetic", "40", "--random_init",
              "--seed", str(seed)])
    out = os.path.join(save, "generation", f"ddpm_MLP_ETTh1_24_{cfg}_{steps}")
    x1 = np.load(os.path.join(out, "x_1.npy"))
    xt = np.load(os.path.join(out, "x_t.npy"))
    lat = np.load(os.path.join(out, "x_t_latent_dec_array.npy"))
    enc = np.load(os.path.join(out, "x_t_latent_enc_array.npy"))
    assert x1.shape == xt.shape == (B, 24, 1) and lat.shape == enc.shape == (B, 64, 30)
    # the rows the (shuffled) loader served, and their embeddings
    ds = SyntheticT2SDataset(40, 24)
    rows = [int(np.argmin(np.abs(ds.samples - x1[i, :, 0][None]).sum(axis=1))) for i in range(B)]
    assert len(set(rows)) == B and np.allclose(ds.samples[rows], x1[:, :, 0], atol=1e-6)
    text = torch.from_numpy(ds.embedding[rows]).float()
    msd, vsd = synth.make_mlp_state_dict(seed), synth.make_vae_state_dict(seed)
    tab = O.ddpm_tables(steps)
    with torch.no_grad():
        z_ref, before = O.vae_encode(vsd, torch.from_numpy(x1[:, :, 0]))
        x = torch.from_numpy(O.device_normal(seed, 0xFFFFFFFF, 0, B, 384)).view(B, 64, 6)
        for j in range(steps):
            t = torch.full((B,), steps - 1 - j, dtype=torch.long)
            u = O.mlp_denoiser_forward(msd, x, t, None)
            c = O.mlp_denoiser_forward(msd, x, t, text)
            x = O.ddpm_p_sample(tab, x, u + cfg * (c - u), t, torch.from_numpy(O.device_normal(seed, j, 0, B, 384)).view(B, 64, 6))
        series, _ = O.vae_decode(vsd, x, 24)
    scale = max(1.0, float(x.abs().max()))
    assert float(np.abs(enc - z_ref.numpy()).max()) < 1e-5
    assert float(np.abs(lat[:, :, :6] - x.numpy()).max()) < 1e-4 * scale and float(np.abs(lat[:, :, 6:]).max()) == 0.0
    assert float(np.abs(xt[:, :, 0] - series.numpy()).max()) < 1e-4 * scale


@pytest.mark.gpu
def test_train_driver_mlp_denoiser_then_infer(tmp_path, monkeypatch):
    """`train.py --denoiser MLP` (reference train.py:16 selects it from the same dict as the DiT): fixed-length L = 24
    data, HIP encoder -> `before` (B,64,6), HIP q_sample / MSE, MLP forward / backward (t2s_mlp_forward / t2s_mlp_backward), the reference checkpoint
    dict -- the first step's loss equals the oracle's on the same rows and draws, the loss goes down, and `infer.py
    --denoiser MLP --checkpoint_id` consumes the checkpoint.  Mixed-length training is refused with a clear message."""
    import infer as idrv
    import train as drv
    assert torch.cuda.is_available()
    monkeypatch.chdir(tmp_path)
    save = str(tmp_path / "results")
    argv = ["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "MLP", "--total_step", "100",
            "--batch_size", "16", "--epochs", "30", "--save_path", save, "--synthetic", "16", "--random_init",
            "--checkpoint_path", "", "--split_train", "--seed", "5"]
    args = drv.get_args(argv)
    assert args.save_path.endswith(os.path.join("checkpoints", "ddpm_MLP_ETTh1_24")) and not args.mix_train
    losses = drv.train(args)
    assert len(losses) == 30 and np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    ck = torch.load(os.path.join(args.save_path, "model_29.pth"), map_location="cpu")
    assert set(ck) == {"model", "optimizer", "epoch", "loss_list"} and ck["epoch"] == 29
    assert "layers.0.norm2.weight" in ck["model"] and sum(1 for k in ck["model"] if k.startswith("encoder.")) == 12
    with pytest.raises(ValueError, match="split_train"):
        drv.train(drv.get_args([a for a in argv if a != "--split_train"]))
    # the checkpoint feeds the sampling driver: infer.py looks under the dataset ROOT name (infer.py:144-146) and loads the
    # whole-module LA-VAE pickle (infer.py:39)
    import shutil
    import types
    from model.pretrained.vqvae import vqvae
    shutil.copytree(args.save_path, os.path.join(save, "checkpoints", "ddpm_MLP_ETTh1"))
    vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    vae.load_state_dict(synth.make_vae_state_dict(5), strict=True)
    os.makedirs("results/saved_pretrained_models/datasetETTh1_epoch2000", exist_ok=True)
    torch.save(vae, "results/saved_pretrained_models/datasetETTh1_epoch2000/final_model.pth")
    idrv.main(["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "MLP", "--total_step", "5", "--cfg_scale", "7",
               "--batch_size", "8", "--save_path", save, "--synthetic", "16", "--seed", "5", "--checkpoint_id", "29"])
    out = os.path.join(save, "generation", "ddpm_MLP_ETTh1_24_7.0_5")
    xt = np.load(os.path.join(out, "x_t.npy"))
    assert xt.shape == (16, 24, 1) and np.isfinite(xt).all()


# ------------------------------------------------------------------ the HIP forward (csrc/t2s_mlp.hip) through the C ABI
def _gpu_mlp(seed=2025):
    from model.denoiser.mlp import MLP
    m = MLP().eval()
    m.load_state_dict(synth.make_mlp_state_dict(seed), strict=True)
    return m.cuda()


@pytest.mark.gpu
def test_hip_mlp_forward_matches_reference_golden(golden_dir):
    """a20: `MLP.forward` on a GPU without autograd = ONE launch of t2s_mlp_forward; against the vectors the reference's own
    mlp.py produced (cond and uncond), fp32 tolerance."""
    g = np.load(os.path.join(golden_dir, "mlp_denoiser.npz"))
    m = _gpu_mlp()
    x = torch.from_numpy(g["x"]).cuda()
    t = torch.tensor([49, 20, 1, 0]).cuda()
    text = synth.make_text_embeddings(5, 4).cuda()
    with torch.no_grad():
        yc, yu = m(x, t, text), m(x, t, None)
    assert "_t2s_packed" in m.__dict__                       # the HIP path ran (its packed weights exist)
    np.testing.assert_allclose(yc.cpu().numpy(), g["cond"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(yu.cpu().numpy(), g["uncond"], atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 32, 257])
def test_hip_mlp_forward_vs_oracle_and_its_own_torch_path(B):
    """Against the CPU oracle on random inputs (long and float t, with and without text), and against the mirror's torch-op
    evaluation on the same GPU (what training runs): one arithmetic, two evaluations; rows are independent of the batch."""
    msd = synth.make_mlp_state_dict(7)
    m = _gpu_mlp(7)
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.randn(B, 64, 6).astype(np.float32))
    text = synth.make_text_embeddings(11, B)
    for t in (torch.from_numpy(rs.randint(0, 50, size=B)), torch.from_numpy(rs.rand(B).astype(np.float32))):
        for tx in (None, text):
            with torch.no_grad():
                y = m(x.cuda(), t.cuda(), None if tx is None else tx.cuda())
                ref = O.mlp_denoiser_forward(msd, x, t, tx)
            scale = max(1.0, float(ref.abs().max()))
            assert float((y.cpu() - ref).abs().max()) < 2e-5 * scale
            xg = x.cuda().requires_grad_(True)                  # autograd wanted -> the torch-op layers
            y_t = m(xg, t.cuda(), None if tx is None else tx.cuda())
            assert y_t.requires_grad and float((y - y_t.detach()).abs().max()) < 2e-5 * scale
            if B > 1:                                           # a row's result does not depend on its batch
                with torch.no_grad():
                    y1 = m(x[:1].cuda(), t[:1].cuda(), None if tx is None else tx[:1].cuda())
                assert torch.equal(y1, y[:1])


@pytest.mark.gpu
def test_hip_mlp_repacks_after_weight_changes_and_checks_extents():
    from t2ms_amd import _lib as L
    m = _gpu_mlp(3)
    x, t = torch.randn(4, 64, 6, device="cuda"), torch.tensor([5, 6, 7, 8], device="cuda")
    text = synth.make_text_embeddings(1, 4).cuda()
    with torch.no_grad():
        y0 = m(x, t, text)
        m.layers[3].mlp[0].weight.mul_(1.5)                     # in place: the version counter moves, the mirror re-packs
        y1 = m(x, t, text)
    xg = x.clone().requires_grad_(True)
    assert not torch.allclose(y0, y1) and float((y1 - m(xg, t, text).detach()).abs().max()) < 2e-5 * max(1.0, float(y1.abs().max()))
    # query / key cannot influence the result (the six keys are one row): scrambling them changes nothing
    with torch.no_grad():
        m.layers[0].cross_attn.query.weight.normal_()
        m.layers[0].cross_attn.key.weight.normal_()
        assert float((m(x, t, text) - m(xg, t, text).detach()).abs().max()) < 2e-5 * max(1.0, float(y1.abs().max()))
    # the C entry points: a short packed buffer, a short input and out == x
    lib = L.lib()
    packed = m.__dict__["_t2s_packed"][2]
    tz = torch.zeros(4, device="cuda")
    fr = torch.pow(10000, torch.linspace(0, 1, 32)).cuda()
    out = torch.empty_like(x)
    # (private hipMallocs: torch's caching allocator hides the end of a small tensor inside a 2 MB segment)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes, hip.hipFree.argtypes = [C.POINTER(C.c_void_p), C.c_size_t], [C.c_void_p]
    short_packed, short_x = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(short_packed), (L.MLP_PACKED_FLOATS - 8) * 4) == 0
    assert hip.hipMalloc(C.byref(short_x), 3 * 384 * 4) == 0
    try:
        assert lib.t2s_mlp_forward(short_packed.value, x.data_ptr(), tz.data_ptr(), fr.data_ptr(), None, out.data_ptr(), 4, None) == -1
        assert b"packed" in lib.t2s_last_error() and b"allocation ends" in lib.t2s_last_error()
        assert lib.t2s_mlp_forward(packed.data_ptr(), short_x.value, tz.data_ptr(), fr.data_ptr(), None, out.data_ptr(), 4, None) == -1
        assert b": x" in lib.t2s_last_error()
    finally:
        hip.hipFree(short_packed)
        hip.hipFree(short_x)
    w = L.MlpWeights()
    assert lib.t2s_mlp_pack(w, packed.data_ptr(), None) == -1 and b"null" in lib.t2s_last_error()
    assert lib.t2s_mlp_forward(packed.data_ptr(), x.data_ptr(), tz.data_ptr(), fr.data_ptr(), None, out.data_ptr(), 0, None) == 0
    L.check(lib.t2s_mlp_forward(packed.data_ptr(), x.data_ptr(), tz.data_ptr(), fr.data_ptr(), None, out.data_ptr(), 4, None))
    xa = x.clone()
    L.check(lib.t2s_mlp_forward(packed.data_ptr(), xa.data_ptr(), tz.data_ptr(), fr.data_ptr(), None, xa.data_ptr(), 4, None))
    torch.cuda.synchronize()
    assert torch.equal(xa, out)


@pytest.mark.gpu
@pytest.mark.parametrize("B,with_text", [(1, True), (5, True), (33, False)])
def test_hip_mlp_backward_vs_autograd(B, with_text):
    """train.py --denoiser MLP: `MLP.forward` under autograd on a GPU = t2s_mlp_forward + t2s_mlp_backward.  All 112 parameter
    gradients and the input gradient against torch autograd through the same mirror's torch-op layers on the CPU in float64;
    cross_attn.query / key get exact zeros (autograd's own are rounding noise); bit-reproducible."""
    import copy
    m = _gpu_mlp(11)
    ref = copy.deepcopy(m).cpu().double()
    rs = np.random.RandomState(100 + B)
    x = torch.from_numpy(rs.randn(B, 64, 6).astype(np.float32))
    t = torch.from_numpy(rs.randint(0, 100, size=B))
    text = synth.make_text_embeddings(13, B) if with_text else None
    wgt = torch.from_numpy(rs.randn(B, 64, 6).astype(np.float32))

    def run_gpu():
        for p in m.parameters():
            p.grad = None
        xg = x.cuda().requires_grad_(True)
        y = m(xg, t.cuda(), None if text is None else text.cuda())
        assert type(y.grad_fn).__name__ == "_MlpFnBackward"
        (y * wgt.cuda()).sum().backward()
        return y.detach(), xg.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    y, dx, g = run_gpu()
    xr = x.double().requires_grad_(True)
    yr = ref(xr, t, None if text is None else text.double())
    (yr * wgt.double()).sum().backward()
    scale = max(1.0, float(yr.detach().abs().max()))
    assert float((y.cpu().double() - yr.detach()).abs().max()) < 2e-5 * scale
    assert float((dx.cpu().double() - xr.grad).abs().max()) < 2e-4 * float(xr.grad.abs().max())
    checked = 0
    # (a layer's mlp2.2.bias adds one value to all 64 channels of a position, which the NEXT layer's LayerNorm removes: its
    # true gradient is zero in layers 0..6 -- hence the absolute floor, relative to the largest gradient of the model)
    floor = 1e-6 * max(float(pr.grad.abs().max()) for pr in ref.parameters() if pr.grad is not None)
    for n, pr in ref.named_parameters():
        if pr.grad is None:
            assert n not in g, n                                   # the never-called modules get no gradient on either side
            continue
        got = g[n].cpu().double()
        if ".cross_attn.query." in n or ".cross_attn.key." in n:
            assert float(got.abs().max()) == 0.0 and float(pr.grad.abs().max()) < 1e-9 * scale, n
            continue
        tol = 2e-4 * float(pr.grad.abs().max()) + floor
        assert float((got - pr.grad).abs().max()) < tol, (n, float((got - pr.grad).abs().max()), tol)
        checked += 1
    assert checked == (8 * 14 if with_text else 8 * 10)
    y2, dx2, g2 = run_gpu()
    assert torch.equal(y, y2) and torch.equal(dx, dx2) and all(torch.equal(g[n], g2[n]) for n in g)
