"""The drop-in claim on FILES THE REFERENCE WROTE.

tests/golden/ref_made/ holds what the reference itself pickled in the build container (tests/golden/gen_golden_r5.py):
`final_model.pth`, the whole-module LA-VAE pickle of pretrained_lavae_unified.py:158 (small hyper-parameters), and
`model_0.pth.gz`, the checkpoint the reference's own train.train(args) wrote after one epoch on the CPU (train.py:134-136).
tests/golden/ref_made.npz holds the reference's outputs for them.  Until round 5 every .pth a test loaded had been written
by this repository's mirrors.

CPU tests: the pickles resolve to the mirrors, keys / dtypes / optimizer layout are what the mirrors expect.
GPU tests: the kernels run on those files -- encode / decode against the stored reference outputs, infer.py loading both
strictly and sampling from them (against the oracle), train.py --checkpoint_path resuming from the reference's checkpoint,
and one resumed optimisation step against the reference's own resumed step."""
import gzip
import os
import shutil

import numpy as np
import pytest
import torch

from oracle import t2s_oracle as O
from t2ms_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MADE = os.path.join(HERE, "golden", "ref_made")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "ref_made.npz"))


def _unpack(tmp_path):
    """Lay the two files out where infer.py:39,48 / train.py:22,42 look for them (relative to the working directory)."""
    vae_dir = tmp_path / "results" / "saved_pretrained_models" / "datasetETTh1_epoch2000"
    ck_dir = tmp_path / "results" / "denoiser_results" / "checkpoints" / "ddpm_DiT_ETTh1"
    os.makedirs(vae_dir)
    os.makedirs(ck_dir)
    shutil.copy(os.path.join(REF_MADE, "final_model.pth"), vae_dir / "final_model.pth")
    with gzip.open(os.path.join(REF_MADE, "model_0.pth.gz"), "rb") as f, open(ck_dir / "model_0.pth", "wb") as o:
        shutil.copyfileobj(f, o)
    return str(vae_dir / "final_model.pth"), str(ck_dir / "model_0.pth")


# ------------------------------------------------------------------------------------------------ CPU: structure
def test_reference_pickles_resolve_to_the_mirrors(tmp_path, g):
    import model.pretrained.vqvae as V
    from model.denoiser.transformer import Transformer
    from t2ms_amd.train import T2SAdamW
    vae_path, ck_path = _unpack(tmp_path)
    vae = torch.load(vae_path, map_location="cpu", weights_only=False)              # infer.py:39
    assert type(vae) is V.vqvae and type(vae.encoder) is V.Encoder and type(vae.decoder) is V.Decoder
    assert type(vae).__module__ == "model.pretrained.vqvae" and V.__file__.startswith(os.path.dirname(HERE))    # the mirror, not the reference
    assert vae.encoder._conv_1.weight.shape == (8, 1, 4) and vae.encoder._pre_vq_conv.weight.shape == (64, 16, 1)   # hidden 16, emb 64
    ck = torch.load(ck_path, map_location="cpu")                                    # train.py:42 (weights_only default)
    assert set(ck) == {"model", "optimizer", "epoch", "loss_list"} and ck["epoch"] == 0
    assert ck["loss_list"] == list(g["ck_loss_list"]) and len(ck["loss_list"]) == 1
    m = Transformer()
    m.encoder = vae.encoder                                                          # infer.py:47 / train.py:30
    assert list(ck["model"].keys()) == list(m.state_dict().keys()) == list(g["ck_model_keys"])
    for k, v in m.state_dict().items():
        assert ck["model"][k].shape == v.shape and ck["model"][k].dtype == v.dtype, k
    m.load_state_dict(ck["model"])                                                   # strict (infer.py:48)
    # optimizer: the reference builds AdamW over model.parameters() (train.py:37): 67 indices, state only where a gradient
    # arrived (the 48 trainable DiT tensors; not pos_embed, not unpatch.*, not the frozen encoder.*)
    names = [n for n, _ in m.named_parameters()]
    pg = ck["optimizer"]["param_groups"][0]
    assert pg["params"] == list(range(len(names))) and sorted(pg.keys()) == list(g["ck_pg_keys"])
    have_state = sorted(ck["optimizer"]["state"].keys())
    assert have_state == list(g["ck_state_idx"]) and len(have_state) == 48
    with_grad = [i for i, n in enumerate(names) if not (n == "pos_embed" or n.startswith("unpatch.") or n.startswith("encoder."))]
    assert have_state == with_grad
    opt = T2SAdamW(m.parameters(), lr=1e-4, weight_decay=0.0)
    opt.load_state_dict(ck["optimizer"])                                             # train.py:44
    assert opt.param_groups[0]["lr"] == float(g["ck_pg_lr"][0]) and tuple(opt.param_groups[0]["betas"]) == tuple(g["ck_pg_betas"])
    for i in have_state:
        st = opt.state[opt.param_groups[0]["params"][i]]
        assert float(st["step"]) == 1.0 and st["exp_avg"].shape == st["exp_avg_sq"].shape == opt.param_groups[0]["params"][i].shape


def test_reference_made_vae_against_the_oracle_on_cpu(tmp_path, g):
    """The oracle's LA-VAE restatement on the state-dict of the reference-pickled module equals the outputs the reference
    computed from that module (pins the oracle at hyper-parameters other than the default ones too)."""
    vae_path, _ = _unpack(tmp_path)
    vsd = torch.load(vae_path, map_location="cpu", weights_only=False).state_dict()
    with torch.no_grad():
        for L in (24, 96):
            z, before = O.vae_encode(vsd, synth.make_series(500 + L, 3, L))
            rec, after = O.vae_decode(vsd, z, L)
            rec_r, _ = O.vae_decode(vsd, synth.make_latents(600 + L, 3), L)
            assert float((z - torch.from_numpy(g[f"vae_z_{L}"])).abs().max()) < 1e-6
            assert float((before - torch.from_numpy(g[f"vae_before_{L}"])).abs().max()) < 1e-6
            assert float((rec - torch.from_numpy(g[f"vae_rec_{L}"])).abs().max()) < 1e-6
            assert float((after - torch.from_numpy(g[f"vae_after_{L}"])).abs().max()) < 1e-6
            assert float((rec_r - torch.from_numpy(g[f"vae_rec_rand_{L}"])).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------------ GPU: the kernels on those files
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_kernels_on_the_reference_pickled_vae(dev, tmp_path, g):
    """torch.load(weights_only=False) -> mirrors -> t2s_vae_encode / t2s_vae_decode within 1e-5 of what the reference
    computed from the module it pickled (infer.py:39-41,73-74,95)."""
    vae_path, _ = _unpack(tmp_path)
    vae = torch.load(vae_path, map_location=torch.device("cpu"), weights_only=False).float().to(dev).eval()
    with torch.no_grad():
        for L in (24, 96):
            z, before = vae.encoder(synth.make_series(500 + L, 3, L).to(dev))
            rec, after = vae.decoder(z, length=L)
            rec_r, after_r = vae.decoder(synth.make_latents(600 + L, 3).to(dev), length=L)
            for name, got in (("z", z), ("before", before), ("rec", rec), ("after", after), ("rec_rand", rec_r), ("after_rand", after_r)):
                ref = torch.from_numpy(g[f"vae_{name}_{L}"])
                assert got.shape == ref.shape, (name, L, got.shape, ref.shape)
                assert float((got.cpu() - ref).abs().max()) < 1e-5, (name, L)
    assert "libt2s_hip.so" in open("/proc/self/maps").read()


@pytest.mark.gpu
def test_infer_loads_the_reference_files_strictly_and_samples_from_them(dev, tmp_path, monkeypatch):
    """infer.py WITHOUT --random_init: the LA-VAE pickle from results/saved_pretrained_models/... (infer.py:39), the checkpoint
    from {save_path}/checkpoints/{backbone}_{denoiser}_{root}/model_{id}.pth, `load_state_dict(...['model'])` strict
    (infer.py:48) -- both files written by the reference -- and the four output files are what the oracle gives with those
    weights on the rows the files name."""
    import infer as drv
    from datafactory.dataset import SyntheticT2SDataset
    vae_path, ck_path = _unpack(tmp_path)
    monkeypatch.chdir(tmp_path)
    seed, L_, n_ds, bs, steps, cfg = 5, 24, 7, 2, 4, 9.0
    save = os.path.join("results", "denoiser_results")
    drv.main(["--dataset_name", f"ETTh1_{L_}", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", str(steps), "--cfg_scale",
              str(cfg), "--batch_size", str(bs), "--save_path", save, "--synthetic", str(n_ds), "--seed", str(seed), "--checkpoint_id",
              "0", "--no_figs"])
    out = os.path.join(save, "generation", f"ddpm_DiT_ETTh1_{L_}_{cfg}_{steps}")
    x1 = np.load(os.path.join(out, "x_1.npy"))[:, :, 0]
    xt = np.load(os.path.join(out, "x_t.npy"))[:, :, 0]
    lat = np.load(os.path.join(out, "x_t_latent_dec_array.npy"))
    enc = np.load(os.path.join(out, "x_t_latent_enc_array.npy"))
    n = (n_ds // bs) * bs
    ds = SyntheticT2SDataset(n_ds, L_)
    rows = [int(np.argmin(np.abs(ds.samples - x1[i][None]).sum(axis=1))) for i in range(n)]
    text = torch.from_numpy(ds.embedding[rows]).float()
    vsd = torch.load(vae_path, map_location="cpu", weights_only=False).state_dict()
    sd = {k: v for k, v in torch.load(ck_path, map_location="cpu")["model"].items() if not k.startswith("encoder.")}
    with torch.no_grad():
        z_ref, _ = O.vae_encode(vsd, torch.from_numpy(x1))
        x_T = torch.from_numpy(O.device_normal(seed, 0xFFFFFFFF, 0, n)).view(n, 64, 30)
        noises = [torch.from_numpy(O.device_normal(seed, j, 0, n)).view(n, 64, 30) for j in range(steps)]
        ref = O.sample_ddpm(sd, x_T, text, steps, cfg, noises)
        series, _ = O.vae_decode(vsd, ref, L_)
    scale = max(1.0, float(ref.abs().max()))
    assert float(np.abs(enc - z_ref.numpy()).max()) < 1e-5
    assert float(np.abs(lat - ref.numpy()).max()) < 1e-4 * scale
    assert float(np.abs(xt - series.reshape(n, L_).numpy()).max()) < 1e-4 * scale


def _resumed_model(dev, tmp_path):
    """train.py:16-47 with the mirrors on the reference's two files."""
    from model.denoiser.transformer import Transformer
    from t2ms_amd.train import T2SAdamW
    vae_path, ck_path = _unpack(tmp_path)
    torch.manual_seed(11)
    m = Transformer().to(dev)
    vae = torch.load(vae_path, map_location=dev, weights_only=False).float().to(dev)
    m.encoder = vae.encoder
    for name, p in m.named_parameters():
        if "encoder" in name:
            p.requires_grad = False
    opt = T2SAdamW(m.parameters(), lr=1e-4, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=2)
    ck = torch.load(ck_path, map_location=dev)
    m.load_state_dict(ck["model"])
    opt.load_state_dict(ck["optimizer"])
    return m, opt, sched, ck


@pytest.mark.gpu
def test_one_resumed_step_equals_the_references_resumed_step(dev, tmp_path, g):
    """train.py:42-47 then ONE step of train.py:118-127 from the reference's checkpoint, on the inputs the reference's own
    resumed step used (x_t, t, emb, noise_gt recorded by the generator): same lr out of the loaded param_groups, loss within
    1e-5, every gradient within 2e-4 max|g| (fp32 mode), Adam step counts 1 -> 2, and the parameters after optimizer.step()
    within 2.5 lr of the reference's (Adam's normalised update is +-lr-sized in its first steps: an element whose gradient
    is within rounding of zero may move by lr in either direction)."""
    from model.backbone.DDPM import DDPM
    m, opt, sched, ck = _resumed_model(dev, tmp_path)
    m.train()
    lr = float(opt.param_groups[0]["lr"])
    assert lr == float(g["rs_lr"])
    x_t, t, emb, noise = (torch.from_numpy(g[k]).to(dev) for k in ("rs_x_t", "rs_t", "rs_emb", "rs_noise"))
    backbone = DDPM(100, dev)
    opt.zero_grad()
    pred = m(input=x_t, t=t, text_input=emb)
    loss = backbone.loss(pred, noise)
    loss.backward()
    assert abs(float(loss) - float(g["rs_loss"])) < 1e-5 * float(g["rs_loss"])
    assert float((pred.detach().flatten()[::53].cpu() - torch.from_numpy(g["rs_pred_sample"])).abs().max()) < 1e-4
    named = dict(m.named_parameters())
    names = list(g["rs_grad_names"])
    assert [n for n, p in m.named_parameters() if p.grad is not None] == names
    for n, ref_norm, ref_max in zip(names, g["rs_grad_norms"], g["rs_grad_maxabs"]):
        got = float(named[n].grad.double().norm())
        assert abs(got - ref_norm) <= 2e-4 * max(ref_norm, ref_max * np.sqrt(named[n].numel()) * 1e-3, 1e-12) + 1e-12, (n, got, ref_norm)
    for n in ("layers.0.adaLN_modulation.1.bias", "layers.3.mlp.fc2.weight", "patch_emb.weight"):
        ref = torch.from_numpy(g["rs_grad__" + n.replace(".", "__")])
        assert float((named[n].grad.cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-12, n
    opt.step()
    sched.step()
    torch.cuda.synchronize()
    steps = [float(opt.state[p]["step"]) for p in opt.param_groups[0]["params"] if p in opt.state and opt.state[p]]
    assert steps == list(g["rs_state_steps"]) and set(steps) == {2.0}
    for n in names:
        ref = torch.from_numpy(g["rs_after__" + n.replace(".", "__")])
        got = named[n].detach().flatten()[::97].cpu()
        assert float((got - ref).abs().max()) <= 2.5 * lr, (n, float((got - ref).abs().max()))
    # ... and most elements agree far closer than the +-lr band (the band is for near-zero gradients only)
    n = "layers.3.mlp.fc2.weight"
    ref = torch.from_numpy(g["rs_after__" + n.replace(".", "__")])
    close = ((named[n].detach().flatten()[::97].cpu() - ref).abs() < 0.05 * lr).float().mean()
    assert float(close) > 0.9, float(close)


@pytest.mark.gpu
def test_train_driver_resumes_from_the_references_checkpoint(dev, tmp_path, monkeypatch, g):
    """`train.py --checkpoint_path <the reference's model_0.pth>` (train.py:42-47): strict load of model and optimizer, the
    loss list carried on, training continues at epoch 1 and the checkpoint it writes at the end holds the reference's first
    loss, its own losses after it, and Adam step counts that continued from 1."""
    import train as drv
    vae_path, ck_path = _unpack(tmp_path)
    monkeypatch.chdir(tmp_path)
    save = os.path.join("results", "denoiser_results")
    drv.train(drv.get_args(["--checkpoint_path", ck_path, "--dataset_name", "ETTh1_24", "--batch_size", "8", "--epochs", "3",
                            "--save_path", save, "--split_train", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100",
                            "--synthetic", "16", "--seed", "3"]))
    out = os.path.join(save, "checkpoints", "ddpm_DiT_ETTh1_24", "model_2.pth")
    ck = torch.load(out, map_location="cpu")
    assert ck["epoch"] == 2
    # 16 synthetic rows / batch 8 = 2 steps per epoch, epochs 1 and 2 ran: the reference's loss + 4 of ours
    assert len(ck["loss_list"]) == 5 and ck["loss_list"][0] == float(g["ck_loss_list"][0])
    assert all(np.isfinite(ck["loss_list"])) and ck["loss_list"][-1] < ck["loss_list"][0]
    steps = {float(st["step"]) for st in ck["optimizer"]["state"].values()}
    assert steps == {5.0}
    assert list(ck["model"].keys()) == list(g["ck_model_keys"])
