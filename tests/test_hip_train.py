"""Training path: DiT backward, MSE, fused AdamW against autograd through the
CPU oracle and the reference-generated fixture tests/golden/train_step.npz.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from oracle import t2s_oracle as O
from t2ms_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(dev, seed=2025):
    from model.denoiser.transformer import Transformer
    m = Transformer()
    m.load_state_dict(synth.make_dit_state_dict(seed), strict=True)
    return m.to(dev).train()


def _oracle_grads(seed, x, t, text, target):
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in synth.make_dit_state_dict(seed).items()}
    pred = O.dit_forward(sd, x, t, text)
    loss = O.mse_loss(pred, target)
    loss.backward()
    return pred.detach(), loss.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}


@pytest.mark.parametrize("with_text", [True, False])
def test_backward_matches_oracle_autograd(dev, with_text):
    B = 3
    x = synth.make_latents(71, B)
    t = torch.tensor([5, 50, 99])
    text = synth.make_text_embeddings(71, B) if with_text else None
    target = synth.make_latents(72, B)
    pred_ref, loss_ref, g_ref = _oracle_grads(2025, x, t, text, target)
    m = _model(dev)
    from t2ms_amd.train import mse_loss
    pred = m(input=x.to(dev), t=t.to(dev), text_input=None if text is None else text.to(dev))
    assert float((pred.detach().cpu() - pred_ref).abs().max()) < 1e-4
    loss = mse_loss(pred, target.to(dev))
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    loss.backward()
    checked = 0
    for name, p in m.named_parameters():
        if name.startswith("unpatch") or name == "pos_embed":
            assert p.grad is None
            continue
        ref = g_ref[name]
        got = p.grad.detach().cpu()
        scale = float(ref.abs().max()) + 1e-12
        err = float((got - ref).abs().max())
        assert err < 2e-4 * scale + 1e-9, (name, err, scale)
        checked += 1
    assert checked == 48


def test_rectified_flow_training_step_matches_oracle(dev):
    """The reference's DEFAULT backbone (train.py:144: flowmatching): t = round(u * total_step) / total_step as a FLOAT
    (train.py:108), x_t = t x_1 + (1 - t) x_0, target x_1 - x_0 (rectified_flow.py:8-16).  The float-t forward and all 48
    gradients against autograd through the oracle."""
    from model.backbone.rectified_flow import RectifiedFlow
    B, total = 4, 100
    z = synth.make_latents(81, B)
    x0 = synth.make_latents(82, B)
    t = torch.round(torch.tensor([0.0, 0.337, 0.5, 1.0]) * total) / total
    text = synth.make_text_embeddings(81, B)
    x_t_ref = O.rf_create_flow(z, t, x0)
    target = z - x0
    pred_ref, loss_ref, g_ref = _oracle_grads(2025, x_t_ref, t, text, target)
    m = _model(dev)
    rf = RectifiedFlow()
    x_t, x_0 = rf.create_flow(z.to(dev), t.to(dev), x_0=x0.to(dev))
    assert float((x_t.cpu() - x_t_ref).abs().max()) < 1e-6 and torch.equal(x_0.cpu(), x0)
    pred = m(input=x_t, t=t.to(dev), text_input=text.to(dev))
    assert float((pred.detach().cpu() - pred_ref).abs().max()) < 1e-4
    loss = rf.loss(pred, target.to(dev))
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    loss.backward()
    checked = 0
    for name, p in m.named_parameters():
        if name.startswith("unpatch") or name == "pos_embed":
            continue
        ref = g_ref[name]
        err = float((p.grad.detach().cpu() - ref).abs().max())
        assert err < 2e-4 * (float(ref.abs().max()) + 1e-12) + 1e-9, (name, err)
        checked += 1
    assert checked == 48


def test_train_driver_default_backbone_then_infer(dev, tmp_path, monkeypatch):
    """train.py with the reference's default --backbone (flowmatching) and --total_step (100), then infer.py with ITS
    defaults (flowmatching, cfg 7, 100 steps -> shortened) from the written checkpoint."""
    import types
    import infer as infer_drv
    import train as drv
    from model.pretrained.vqvae import vqvae
    monkeypatch.chdir(tmp_path)
    vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    vae.load_state_dict(synth.make_vae_state_dict(2025), strict=True)
    os.makedirs("results/saved_pretrained_models/datasetETTh1_epoch2000")
    torch.save(vae, "results/saved_pretrained_models/datasetETTh1_epoch2000/final_model.pth")
    save = str(tmp_path / "results" / "denoiser_results")
    argv = ["--dataset_name", "ETTh1", "--batch_size", "12", "--epochs", "4", "--save_path", save, "--synthetic", "24",
            "--checkpoint_path", "", "--seed", "7"]
    monkeypatch.setattr("sys.argv", ["train.py"] + argv)
    args = drv.get_args(argv)
    assert args.backbone == "flowmatching" and args.total_step == 100 and args.denoiser == "DiT"
    losses = drv.train(args)
    assert np.isfinite(losses).all() and len(losses) >= 8
    assert os.path.exists(os.path.join(save, "checkpoints", "flowmatching_DiT_ETTh1", "model_3.pth"))
    infer_drv.main(["--dataset_name", "ETTh1_48", "--total_step", "5", "--batch_size", "4", "--save_path", save,
                    "--synthetic", "9", "--checkpoint_id", "3", "--seed", "3"])
    gen = np.load(os.path.join(save, "generation", "flowmatching_DiT_ETTh1_48_7_5", "x_t.npy"))
    assert gen.shape == (8, 48, 1) and np.isfinite(gen).all()


def test_train_driver_resident_batches_equal_the_loader_pass(dev, tmp_path, monkeypatch):
    """train.py's default data path gathers index batches from device-resident latent / embedding tables
    (datafactory.epoch_index_batches: the order a pass over the DataLoader draws, no per-row __getitem__ / collate);
    `--loader_batches` walks the DataLoader as the reference does (train.py:52-95, mix-train: three length groups per
    batch).  Same rows, same draws: the loss lists and the final weights must be IDENTICAL, and --max_steps stops a run."""
    import train as drv
    monkeypatch.chdir(tmp_path)
    runs = {}
    for tag, extra in (("resident", []), ("loader", ["--loader_batches"])):
        save = str(tmp_path / tag)
        argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--batch_size", "10", "--epochs", "3", "--save_path", save,
                "--synthetic", "9", "--random_init", "--checkpoint_path", "", "--seed", "5", "--bf16"] + extra
        losses = drv.train(drv.get_args(argv))
        ck = torch.load(os.path.join(save, "checkpoints", "ddpm_DiT_ETTh1", "model_2.pth"), map_location="cpu")
        assert ck["loss_list"] == losses
        runs[tag] = (losses, ck["model"])
    (la, wa), (lb, wb) = runs["resident"], runs["loader"]
    assert len(la) == len(lb) >= 6 and la == lb, (la, lb)            # 3 epochs x 2 batches x up to 3 length groups
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k
    args = drv.get_args(["--dataset_name", "ETTh1", "--backbone", "ddpm", "--batch_size", "10", "--epochs", "3", "--save_path",
                         str(tmp_path / "short"), "--synthetic", "9", "--random_init", "--checkpoint_path", "", "--seed", "5",
                         "--bf16", "--max_steps", "4"])
    ticks = []
    args.on_step = lambda n, rows, model: ticks.append((n, rows))
    short = drv.train(args)
    assert short == la[:4] and [n for n, _ in ticks] == [1, 2, 3, 4] and all(r > 0 for _, r in ticks)


def test_train_driver_first_steps_equal_the_oracle(dev, tmp_path, monkeypatch):
    """train.py itself against the CPU oracle, fp32 arithmetic: the losses of the first optimisation steps of a split-train
    DDPM run must be what the oracle's restatement of train.py:103-127 gives on the rows and CFG coins the driver used
    (recorded at its `train_step` calls; that the resident order IS the loader's order is
    test_train_driver_resident_batches_equal_the_loader_pass) -- latents from the (cached) frozen encoder, per-row
    t = floor(u T) with u from the library's Philox uniform stream, targets from its normal stream, AdamW + OneCycleLR
    between the steps (oracle side: torch autograd, torch.optim.AdamW)."""
    import train as drv
    from datafactory.dataset import SyntheticT2SDataset
    monkeypatch.chdir(tmp_path)
    seed, B, n_ds, T, n_steps = 13, 6, 12, 100, 3
    argv = ["--dataset_name", "ETTh1_96", "--backbone", "ddpm", "--batch_size", str(B), "--epochs", "2", "--save_path",
            str(tmp_path / "res"), "--synthetic", str(n_ds), "--random_init", "--checkpoint_path", "", "--seed", str(seed),
            "--split_train", "--max_steps", str(n_steps)]
    calls, real = [], drv.train_step

    def spy(model, backbone, opt, dist, args, x_1, emb, device, rank, world, latents=None, idx=None, step_no=0,
            emb_table=None, drop_text=None):
        calls.append((idx.cpu().clone(), bool(drop_text), int(step_no), float(opt.param_groups[0]["lr"])))
        return real(model, backbone, opt, dist, args, x_1, emb, device, rank, world, latents, idx, step_no, emb_table, drop_text)

    monkeypatch.setattr(drv, "train_step", spy)
    losses = drv.train(drv.get_args(argv))
    assert len(losses) == n_steps == len(calls) and [c[2] for c in calls] == [0, 1, 2]
    assert sorted(torch.cat([calls[0][0], calls[1][0]]).tolist()) == list(range(n_ds))      # epoch 0 = a permutation
    # ---- the oracle's run of the same three steps
    ds = SyntheticT2SDataset(n_ds, 96)
    sd = {k: v.clone().requires_grad_(not k.startswith(("pos_embed", "unpatch."))) for k, v in synth.make_dit_state_dict(seed).items()}
    vsd = synth.make_vae_state_dict(seed)
    opt = torch.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=1e-4, weight_decay=0.0)
    tab = O.ddpm_tables(T)
    want = []
    for idx, coin, step_no, lr in calls:
        x1 = torch.from_numpy(ds.samples[idx.numpy()]).float()
        emb = torch.from_numpy(ds.embedding[idx.numpy()]).float()
        with torch.no_grad():
            z, _ = O.vae_encode(vsd, x1)
        u = torch.from_numpy(O.device_uniform(seed ^ drv.TIME_KEY, step_no, 0, B, 1)).view(B)
        t = torch.floor(u * T).long()
        eps = torch.from_numpy(O.device_normal(seed ^ 0x7261696E, step_no, 0, B)).view(B, 64, 30)
        pred = O.dit_forward(sd, O.ddpm_q_sample(tab, z, t, eps), t, None if coin else emb)
        loss = torch.nn.functional.mse_loss(pred, eps)
        opt.zero_grad()
        loss.backward()
        opt.param_groups[0]["lr"] = lr              # the OneCycleLR value the driver stepped with (4e-6 at the start)
        opt.step()
        want.append(float(loss.detach()))
    assert 0 < calls[0][3] < 1e-4
    for a, b in zip(losses, want):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (losses, want)


def test_train_step_reference_fixture(golden_dir, dev):
    """Fixture (9): loss and the 48 per-parameter gradient norms produced by the reference itself."""
    g = np.load(os.path.join(golden_dir, "train_step.npz"))
    from model.backbone.DDPM import DDPM
    m = _model(dev)
    d100 = DDPM(100, dev)
    x1 = synth.make_latents(555, 4).to(dev)
    tt = torch.tensor([3, 50, 77, 99], device=dev)
    eps = synth.make_latents(556, 4).to(dev)
    xt, _ = d100.q_sample(x1, tt, eps)
    pred = m(input=xt, t=tt, text_input=synth.make_text_embeddings(555, 4).to(dev))
    loss = d100.loss(pred, eps)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-5)
    n = 0
    for name, p in m.named_parameters():
        key = "gn_" + name.replace(".", "__")
        if key in g.files:
            np.testing.assert_allclose(p.grad.norm().item(), g[key], rtol=5e-4, atol=1e-7, err_msg=name)
            n += 1
    assert n == 48
    got = m.layers[0].attn.qkv.weight.grad[::16, ::8].cpu().numpy()
    np.testing.assert_allclose(got, g["grad_qkv0_sample"], atol=2e-6, rtol=2e-3)


def test_fused_adamw_matches_torch(dev):
    from t2ms_amd.train import T2SAdamW
    rs = np.random.RandomState(4)
    p0 = torch.from_numpy(rs.randn(1000).astype(np.float32))
    pa = torch.nn.Parameter(p0.clone())
    pb = torch.nn.Parameter(p0.clone().to(dev))
    oa = torch.optim.AdamW([pa], lr=1e-2, weight_decay=0.1)
    ob = T2SAdamW([pb], lr=1e-2, weight_decay=0.1)
    for _ in range(5):
        g = torch.from_numpy(rs.randn(1000).astype(np.float32))
        pa.grad, pb.grad = g.clone(), g.clone().to(dev)
        oa.step()
        ob.step()
    assert float((pb.detach().cpu() - pa.detach()).abs().max()) < 1e-6
    sa, sb = oa.state_dict(), ob.state_dict()
    assert set(sa["state"][0]) == set(sb["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ob2 = T2SAdamW([pb], lr=1e-2, weight_decay=0.1)
    ob2.load_state_dict(sb)          # round trip


def test_resumed_optimizer_keeps_step_on_the_host(dev, tmp_path):
    """train.py:44 resumes with torch.load(..., map_location=device): Optimizer.load_state_dict leaves `step` where the
    load put it, i.e. on the GPU, and every later step would bump and .item() 48 device scalars (48 host syncs per
    step).  T2SAdamW.load_state_dict brings it back to the CPU; the update continues exactly where the checkpoint's
    torch.optim.AdamW left off."""
    from t2ms_amd.train import T2SAdamW
    rs = np.random.RandomState(7)
    ps = [torch.nn.Parameter(torch.from_numpy(rs.randn(n).astype(np.float32)).to(dev)) for n in (300, 4100)]
    ref = torch.optim.AdamW(ps, lr=1e-2, weight_decay=0.0)
    grads = [[torch.from_numpy(rs.randn(p.numel()).astype(np.float32)).to(dev) for p in ps] for _ in range(4)]
    for it in range(2):
        for p, g in zip(ps, grads[it]):
            p.grad = g.clone()
        ref.step()
    f = tmp_path / "ck.pth"
    torch.save(dict(optimizer=ref.state_dict(), params=[p.detach().clone() for p in ps]), f)
    ck = torch.load(f, map_location=dev)
    assert ck["optimizer"]["state"][0]["step"].device.type == "cuda"        # what the resume path hands us
    qs = [torch.nn.Parameter(t.clone()) for t in ck["params"]]
    ours = T2SAdamW(qs, lr=1e-2, weight_decay=0.0)
    ours.load_state_dict(ck["optimizer"])
    for q in qs:
        st = ours.state[q]
        assert st["step"].device.type == "cpu" and float(st["step"]) == 2.0
        assert st["exp_avg"].device.type == "cuda"
    for it in (2, 3):
        for p, q, g in zip(ps, qs, grads[it]):
            p.grad, q.grad = g.clone(), g.clone()
        ref.step()
        ours.step()
    for p, q in zip(ps, qs):
        assert ours.state[q]["step"].device.type == "cpu" and float(ours.state[q]["step"]) == 4.0
        assert float((p.detach() - q.detach()).abs().max()) < 1e-6


def test_few_training_steps_reduce_loss_and_refresh_weights(dev):
    """train.py:101-127 in miniature: q_sample -> forward -> mse -> backward -> AdamW; the sampler's
    packed weights follow the optimizer's in-place updates."""
    from model.backbone.DDPM import DDPM
    from t2ms_amd.train import T2SAdamW
    m = _model(dev, seed=9)
    opt = T2SAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    x1 = synth.make_latents(1, 8).to(dev) * 0.5
    text = synth.make_text_embeddings(1, 8).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    losses = []
    for it in range(12):
        t = torch.randint(0, 100, (8,), generator=g).to(dev)
        eps = torch.randn(8, 64, 30, generator=g).to(dev)
        xt, _ = ddpm.q_sample(x1, t, eps)
        opt.zero_grad()
        loss = ddpm.loss(m(input=xt, t=t, text_input=text), eps)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < np.mean(losses[:3])
    # inference forward after training uses the UPDATED weights (packed-weight cache invalidated)
    with torch.no_grad():
        y = m(input=x1, t=torch.zeros(8, dtype=torch.long, device=dev), text_input=text)
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref = O.dit_forward(sd, x1.cpu(), torch.zeros(8, dtype=torch.long), text.cpu())
    assert float((y.cpu() - ref).abs().max()) < 1e-4


def test_train_driver_checkpoint_and_resume(dev, tmp_path, monkeypatch):
    """train.py drop-in: reference flags / path derivation / checkpoint dict (train.py:94,134-136,157),
    mix-train over three lengths, then resume from the written checkpoint."""
    import train as drv
    monkeypatch.chdir(tmp_path)
    save = str(tmp_path / "results")
    argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100",
            "--batch_size", "12", "--epochs", "2", "--save_path", save, "--synthetic", "16", "--random_init",
            "--checkpoint_path", ""]
    monkeypatch.setattr("sys.argv", ["train.py"] + argv)
    args = drv.get_args(argv)
    assert args.save_path.endswith(os.path.join("checkpoints", "ddpm_DiT_ETTh1")) and args.mix_train
    losses = drv.train(args)
    ck_path = os.path.join(args.save_path, "model_1.pth")
    ck = torch.load(ck_path, map_location="cpu")
    assert set(ck) == {"model", "optimizer", "epoch", "loss_list"} and ck["epoch"] == 1
    assert len(ck["loss_list"]) == len(losses) > 0 and np.isfinite(losses).all()
    assert sum(1 for k in ck["model"] if k.startswith("encoder.")) == 12 and "layers.3.mlp.fc2.weight" in ck["model"]
    st = ck["optimizer"]["state"]
    assert len(st) == 48 and set(next(iter(st.values()))) == {"step", "exp_avg", "exp_avg_sq"}
    # resume for one more epoch
    args2 = drv.get_args(argv[:-1] + [ck_path] + ["--epochs", "3"])
    losses2 = drv.train(args2)
    assert len(losses2) > len(losses)


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("with_text", [True, False])
def test_bf16_backward_close_to_fp32_oracle(dev, with_text):
    """BASELINE config 4 arithmetic (bf16 MFMA operands / saved activations, fp32 accumulate, fp32
    residual stream and statistics): every gradient tensor stays within bf16 rounding of autograd
    through the fp32 oracle -- relative L2 error <= 1e-2 (measured 2e-4 .. 3e-3) and cosine >= 0.9999 per tensor."""
    B = 3
    x = synth.make_latents(71, B)
    t = torch.tensor([5, 50, 99])
    text = synth.make_text_embeddings(71, B) if with_text else None
    target = synth.make_latents(72, B)
    pred_ref, loss_ref, g_ref = _oracle_grads(2025, x, t, text, target)
    m = _model(dev).set_train_dtype("bf16")
    from t2ms_amd.train import mse_loss
    pred = m(input=x.to(dev), t=t.to(dev), text_input=None if text is None else text.to(dev))
    assert _rel(pred.detach().cpu(), pred_ref) < 5e-3
    loss = mse_loss(pred, target.to(dev))
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-3)
    loss.backward()
    checked, worst = 0, ("", 0.0)
    for name, p in m.named_parameters():
        if name.startswith("unpatch") or name == "pos_embed":
            assert p.grad is None
            continue
        ref = g_ref[name].flatten().double()
        got = p.grad.detach().cpu().flatten().double()
        assert torch.isfinite(got).all(), name
        rel = _rel(got, ref)
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        if rel > worst[1]:
            worst = (name, rel)
        assert rel < 1e-2 and cos > 0.9999, (name, rel, cos)
        checked += 1
    assert checked == 48, worst
    # switching back to fp32 on the same handle restores the exact path
    m.set_train_dtype("f32")
    m.zero_grad()
    pred32 = m(input=x.to(dev), t=t.to(dev), text_input=None if text is None else text.to(dev))
    assert float((pred32.detach().cpu() - pred_ref).abs().max()) < 1e-4


def test_bf16_training_reduces_loss(dev):
    from model.backbone.DDPM import DDPM
    from t2ms_amd.train import T2SAdamW
    m = _model(dev, seed=9).set_train_dtype("bf16")
    opt = T2SAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    x1 = synth.make_latents(1, 8).to(dev) * 0.5
    text = synth.make_text_embeddings(1, 8).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    losses = []
    for it in range(12):
        t = torch.randint(0, 100, (8,), generator=g).to(dev)
        eps = torch.randn(8, 64, 30, generator=g).to(dev)
        xt, _ = ddpm.q_sample(x1, t, eps)
        opt.zero_grad()
        loss = ddpm.loss(m(input=xt, t=t, text_input=text), eps)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < np.mean(losses[:3])


def test_bf16_attention_forward_and_stale_reference_branch(dev):
    """attn16_fwd_kernel through t2s_attn_fwd_bf16: output and log2-domain LSE against an fp64 softmax of
    the bf16-rounded operands, with spikes that make the sticky reference (row max of the FIRST key
    block) stale by far more than 2^40 at early and late key blocks."""
    from t2ms_amd import _lib as L
    rs = np.random.RandomState(33)
    n_seq = 2
    BH = n_seq * 4
    q, k, v = (torch.from_numpy(rs.randn(BH, 480, 32).astype(np.float32)) for _ in range(3))
    k[:, 333] = q[:, 100] * 5.0
    k[:, 410] = q[:, 200] * 12.0                       # log2-domain jump of ~100 at key block 12
    k[:, 0:32] = -q[:, 7:8] * 3.0 + 0.01 * k[:, 0:32]  # first block hugely negative for query 7 ...
    k[:, 448] = q[:, 7] * 10.0                         # ... and its real maximum in block 14
    # the kernels take q PRE-SCALED by log2(e)/sqrt(32) before the bf16 rounding (t2s_bf16.h ATT_QS; the wrapper applies it
    # as the qkv GEMM's epilogue does): the reference uses exactly those operands, scores in the log2 domain
    QS = float(np.float32(0.17677669529663687) * np.float32(1.4426950408889634))
    qb = (q * QS).to(torch.bfloat16).double()
    kb, vb = (t.to(torch.bfloat16).double() for t in (k, v))
    s = (qb @ kb.transpose(-1, -2)) * np.log(2.0)
    ref = torch.softmax(s, dim=-1) @ vb
    lse_ref = torch.logsumexp(s, dim=-1) / np.log(2.0)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    od = torch.empty(n_seq * 480, 128, device=dev)
    lsed = torch.empty(BH, 480, device=dev)
    L.check(L.lib().t2s_attn_fwd_bf16(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), od.data_ptr(), lsed.data_ptr(),
                                      n_seq, L.stream_ptr(dev)), "t2s_attn_fwd_bf16")
    o = od.cpu().reshape(n_seq, 480, 4, 32).permute(0, 2, 1, 3).reshape(BH, 480, 32).double()
    assert torch.isfinite(o).all()
    # P is rounded to bf16 before P.V and O is stored as bf16: ~2^-8 relative
    assert float((o - ref).abs().max()) < 3e-2 and _rel(o, ref) < 6e-3
    assert float((lsed.cpu().double() - lse_ref).abs().max()) < 2e-3


def test_bf16_full_batch_properties(dev):
    """BASELINE configs[3] per-GPU shape (B=1152, bf16): size-independent properties instead of an oracle run --
    the forward rows equal the SAME rows in a 4-row batch bitwise (rows are independent end to end, tile position in
    the batch does not change a row's arithmetic), every gradient is finite, and ALL 48 gradients are
    bit-reproducible from run to run (fixed reduction order; include/t2s.h t2s_dit_train_backward)."""
    from t2ms_amd.train import _trainable, mse_loss
    B = 1152
    x = synth.make_latents(5, B).to(dev)
    text = synth.make_text_embeddings(5, B).to(dev)
    t = (torch.arange(B) % 100).to(dev)
    target = synth.make_latents(6, B).to(dev)
    m = _model(dev).set_train_dtype("bf16")
    rows = [0, 1, 575, 1151]
    with torch.enable_grad():
        small = m(input=x[rows].contiguous(), t=t[rows].contiguous(), text_input=text[rows].contiguous()).detach().clone()
        runs = []
        for _ in range(2):
            m.zero_grad()
            pred = m(input=x, t=t, text_input=text)
            loss = mse_loss(pred, target)
            loss.backward()
            torch.cuda.synchronize()
            runs.append((pred.detach().clone(), float(loss), [p.grad.detach().clone() for p in _trainable(m)]))
    assert torch.equal(runs[0][0][rows], small), "a row's forward depends on the batch it sits in"
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    names = ["conv_w", "conv_b", "patch_w", "patch_b", "ln_w", "ln_b", "out_w", "out_b"] + \
            [f"blk{i}.{n}" for i in range(4) for n in ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w",
                                                        "fc2_b", "ada_w", "ada_b")]
    for name, g0, g1 in zip(names, runs[0][2], runs[1][2]):
        assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0, name
        # block weights AND the final-layer / patchify tail gradients (per-workgroup partial rows reduced in workgroup
        # order since round 2): every reduction has a fixed order
        assert torch.equal(g0, g1), f"{name}: gradient differs between two identical runs"


def test_backward_fails_loudly_after_a_second_grad_forward(dev):
    """The saved activations live in the model's one handle: a second grad-mode forward before the backward of the
    first must raise instead of returning the gradients of the wrong pass."""
    from t2ms_amd._lib import T2SError
    from t2ms_amd.train import mse_loss
    m = _model(dev)
    x = synth.make_latents(1, 2).to(dev)
    t = torch.tensor([3, 4], device=dev)
    p1 = m(input=x, t=t, text_input=None)
    p2 = m(input=x * 2, t=t, text_input=None)
    with pytest.raises(T2SError, match="another grad-mode forward"):
        mse_loss(p1, x).backward()
    m.zero_grad()
    mse_loss(p2, x).backward()                # the most recent forward is still differentiable
    assert m.layers[0].attn.qkv.weight.grad is not None
    # accumulation over two forward/backward pairs adds up (the persistent bucket is not overwritten in place)
    g1 = m.layers[0].attn.qkv.weight.grad.clone()
    mse_loss(m(input=x * 2, t=t, text_input=None), x).backward()
    torch.testing.assert_close(m.layers[0].attn.qkv.weight.grad, 2 * g1, rtol=1e-5, atol=1e-8)


def test_model_pickles_after_a_training_step(dev):
    """The handle, the packed weights and the persistent gradient bucket (device pointers in ctypes structs) are process
    state, not model state: a module that has trained still pickles / deep-copies (whole-module checkpoints, infer.py:39
    style) and the copy trains on."""
    import copy
    import pickle
    from t2ms_amd.train import mse_loss
    m = _model(dev)
    x = synth.make_latents(1, 2).to(dev)
    t = torch.tensor([3, 4], device=dev)
    mse_loss(m(input=x, t=t, text_input=None), x).backward()
    blob = pickle.dumps(m)
    m2 = pickle.loads(blob)
    m3 = copy.deepcopy(m)
    for mm in (m2, m3):
        mm.zero_grad()
        mse_loss(mm(input=x, t=t, text_input=None), x).backward()
        torch.testing.assert_close(mm.layers[0].attn.qkv.weight.grad, m.layers[0].attn.qkv.weight.grad, rtol=1e-5, atol=1e-9)


def test_drop_in_chain_vae_pickle_train_infer_metrics(dev, tmp_path, monkeypatch, capsys):
    """The reference's whole flow through the drop-in entry points, nothing seeded behind its back: a whole-module LA-VAE
    pickle where train.py / infer.py look for it (train.py:22,156; infer.py:39), train.py from `initialize_weights` in bf16
    -> checkpoint dict (train.py:94) -> infer.py loads THAT checkpoint by --checkpoint_id (infer.py:48,145) and writes the four
    .npy files -> the GPU metrics CLI reads them (what evaluation.py:285-297 reads)."""
    import types
    import infer as infer_drv
    import train as train_drv
    from model.pretrained.vqvae import vqvae
    from t2ms_amd import metrics
    monkeypatch.chdir(tmp_path)
    vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    vae.load_state_dict(synth.make_vae_state_dict(2025), strict=True)
    os.makedirs("results/saved_pretrained_models/datasetETTh1_epoch2000")
    torch.save(vae, "results/saved_pretrained_models/datasetETTh1_epoch2000/final_model.pth")
    save = str(tmp_path / "results" / "denoiser_results")
    targv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100", "--batch_size", "12",
             "--epochs", "3", "--save_path", save, "--synthetic", "24", "--checkpoint_path", "", "--bf16", "--seed", "5"]
    monkeypatch.setattr("sys.argv", ["train.py"] + targv)
    losses = train_drv.train(train_drv.get_args(targv))
    assert np.isfinite(losses).all() and len(losses) >= 6
    ck = os.path.join(save, "checkpoints", "ddpm_DiT_ETTh1", "model_2.pth")
    assert os.path.exists(ck)
    trained = torch.load(ck, map_location="cpu")["model"]
    assert float(trained["layers.0.adaLN_modulation.1.weight"].abs().max()) > 0.0      # adaLN-Zero has left zero: it trained
    infer_drv.main(["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "4", "--cfg_scale", "7",
                    "--batch_size", "4", "--save_path", save, "--synthetic", "9", "--checkpoint_id", "2", "--seed", "3"])
    out = os.path.join(save, "generation", "ddpm_DiT_ETTh1_24_7.0_4")
    gen = np.load(os.path.join(out, "x_t.npy"))
    assert gen.shape == (8, 24, 1) and np.isfinite(gen).all()
    capsys.readouterr()
    metrics.main([out])
    line = capsys.readouterr().out
    assert "MSE" in line and "WAPE" in line and "DTW" in line and "nan" not in line.lower(), line
    assert "C-FID" in line and "200 iterations" in line, line      # TS2Vec trained + encoded on the GPU: C-FID from the .npy files alone


def test_unfrozen_encoder_gradients_match_oracle(dev):
    """train.py:31-33 with `usepretrainedvae` false: the LA-VAE encoder trains jointly.  The encoder runs forward AND backward
    in the HIP kernels (t2s_vae_encode / t2s_vae_encode_backward behind an autograd.Function; round 5 -- until then torch conv
    ops under autograd), q_sample as torch glue, and the DiT returns the gradient of its input latent
    (t2s_dit_train_input_grad, patchify backwards): all 12 encoder gradients and the 48 DiT gradients against autograd
    through the oracle (encoder -> q_sample -> DiT -> MSE), 2e-4 of each tensor's largest gradient."""
    import types
    from model.pretrained.vqvae import vqvae
    from t2ms_amd.train import mse_loss
    B, Ls, T = 3, 96, 100
    xs = synth.make_series(91, B, Ls)
    text = synth.make_text_embeddings(91, B)
    t = torch.tensor([3, 41, 97])
    noise = synth.make_latents(92, B)
    # oracle
    sd = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in synth.make_dit_state_dict(2025).items()}
    vsd = {k: v.clone().requires_grad_(k.startswith("encoder.")) for k, v in synth.make_vae_state_dict(2025).items()}
    tab = O.ddpm_tables(T)
    z_ref, _ = O.vae_encode(vsd, xs)
    xt_ref = O.ddpm_q_sample(tab, z_ref, t, noise)
    loss_ref = O.mse_loss(O.dit_forward(sd, xt_ref, t, text), noise)
    loss_ref.backward()
    # HIP DiT + autograd encoder
    m = _model(dev)
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    v.load_state_dict(synth.make_vae_state_dict(2025), strict=True)
    m.encoder = v.encoder.to(dev)
    z, _ = m.encoder(xs.to(dev))
    assert z.requires_grad and float((z.detach().cpu() - z_ref.detach()).abs().max()) < 1e-5
    assert type(z.grad_fn).__name__ == "_EncodeFnBackward"          # the HIP pair, not torch conv ops
    ab = tab["alpha_bar"].to(dev).gather(-1, t.to(dev)).reshape(-1, 1, 1)
    x_t = ab ** 0.5 * z + (1 - ab) ** 0.5 * noise.to(dev)
    loss = mse_loss(m(input=x_t, t=t.to(dev), text_input=text.to(dev)), noise.to(dev))
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-5)
    loss.backward()
    checked = 0
    for name, p in m.named_parameters():
        if name.startswith("unpatch") or name == "pos_embed":
            continue
        ref = (vsd[name] if name.startswith("encoder.") else sd[name]).grad
        got = p.grad.detach().cpu()
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) < 2e-4 * scale + 1e-9, (name, float((got - ref).abs().max()), scale)
        checked += 1
    assert checked == 48 + 12
    # with the encoder frozen the forward is the HIP kernel again and the DiT is not asked for an input gradient
    for p in m.encoder.parameters():
        p.requires_grad = False
    z2, _ = m.encoder(xs.to(dev))
    assert not z2.requires_grad and float((z2 - z.detach()).abs().max()) < 1e-5


@pytest.mark.parametrize("Ls,B", [(24, 5), (48, 2), (96, 3), (128, 1)])
def test_encoder_backward_kernel_vs_oracle_autograd(dev, Ls, B, monkeypatch):
    """t2s_vae_encode_backward (vqvae.py:57-71 backwards) on its own: loss = <z, G> + <before, H> with random G, H, so both
    outputs carry a gradient.  All 12 encoder.* gradients against autograd through the oracle (2e-4 of each tensor's largest
    gradient; measured ~1e-6), bit-reproducible from run to run, equal (to rounding) to the torch-op path it replaced, and
    fresh weights after an in-place update are picked up without rebuilding the handle (t2s_vae_update_weights)."""
    import types
    from model.pretrained.vqvae import vqvae
    xs = synth.make_series(300 + Ls, B, Ls)
    rs = np.random.RandomState(Ls)
    G = torch.from_numpy(rs.randn(B, 64, 30).astype(np.float32))
    Hh = torch.from_numpy(rs.randn(B, 64, Ls // 4).astype(np.float32))
    vsd = {k: v.clone().requires_grad_(k.startswith("encoder.")) for k, v in synth.make_vae_state_dict(2025).items()}
    z_ref, before_ref = O.vae_encode(vsd, xs)
    ((z_ref * G).sum() + (before_ref * Hh).sum()).backward()
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    v.load_state_dict(synth.make_vae_state_dict(2025), strict=True)
    enc = v.encoder.to(dev)

    def run():
        enc.zero_grad(set_to_none=True)
        z, before = enc(xs.to(dev))
        ((z * G.to(dev)).sum() + (before * Hh.to(dev)).sum()).backward()
        return z, {n: p.grad.detach().clone() for n, p in enc.named_parameters()}

    z, g1 = run()
    assert type(z.grad_fn).__name__ == "_EncodeFnBackward"
    _, g2 = run()
    assert len(g1) == 12
    for n, got in g1.items():
        ref = vsd["encoder." + n].grad
        scale = float(ref.abs().max()) + 1e-12
        assert got.shape == ref.shape
        assert float((got.cpu() - ref).abs().max()) < 2e-4 * scale, (n, float((got.cpu() - ref).abs().max()), scale)
        assert torch.equal(got, g2[n]), n                                   # fixed summation order
    h_before = enc.__dict__["_t2s_h"]
    monkeypatch.setenv("T2S_ENCODER_TORCH_AUTOGRAD", "1")                    # the torch-op path this kernel replaced
    zt, g3 = run()
    assert type(zt.grad_fn).__name__ != "_EncodeFnBackward"
    for n in g1:
        scale = float(g3[n].abs().max()) + 1e-12
        assert float((g1[n] - g3[n]).abs().max()) < 2e-4 * scale, n
    monkeypatch.delenv("T2S_ENCODER_TORCH_AUTOGRAD")
    # an optimizer-style in-place update: same handle, new contents
    with torch.no_grad():
        for p in enc.parameters():
            p.mul_(1.01)
    z2, g4 = run()
    assert enc.__dict__["_t2s_h"] is h_before
    vsd2 = {k: (v.detach() * 1.01 if k.startswith("encoder.") else v.detach()).clone().requires_grad_(k.startswith("encoder."))
            for k, v in vsd.items()}
    z2_ref, b2_ref = O.vae_encode(vsd2, xs)
    ((z2_ref * G).sum() + (b2_ref * Hh).sum()).backward()
    assert float((z2.detach().cpu() - z2_ref.detach()).abs().max()) < 1e-5
    for n, got in g4.items():
        ref = vsd2["encoder." + n].grad
        assert float((got.cpu() - ref).abs().max()) < 2e-4 * (float(ref.abs().max()) + 1e-12), n


def test_encoder_backward_refuses_what_it_does_not_cover(dev):
    """Non-default LA-VAE shapes keep the labelled torch-op path in the mirror; the C entry itself refuses them loudly."""
    import ctypes as C
    import types
    from model.pretrained.vqvae import vqvae
    from t2ms_amd import _lib as L
    v = vqvae(types.SimpleNamespace(block_hidden_size=16, num_residual_layers=2, res_hidden_size=32, embedding_dim=64)).to(dev)
    x = synth.make_series(1, 2, 24).to(dev)
    z, _ = v.encoder(x)
    assert z.requires_grad and type(z.grad_fn).__name__ != "_EncodeFnBackward"
    z.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in v.encoder.parameters())
    h = v.encoder._handle(dev)
    g = L.VaeEncGrads()
    rc = L.lib().t2s_vae_encode_backward(h, x.data_ptr(), z.detach().data_ptr(), None, C.byref(g), 2, 24, None)
    assert rc == -1 and "unsupported" in L.lib().t2s_last_error().decode()
    # a series that itself asks for a gradient (never in train.py, where it is data) is not silently given none: the default
    # shape then takes the torch-op forward too
    vd = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64)).to(dev)
    xg = synth.make_series(2, 2, 24).to(dev).requires_grad_(True)
    zg, _ = vd.encoder(xg)
    assert type(zg.grad_fn).__name__ != "_EncodeFnBackward"
    zg.sum().backward()
    assert xg.grad is not None and float(xg.grad.abs().max()) > 0
    zd, _ = vd.encoder(xg.detach())
    assert type(zd.grad_fn).__name__ == "_EncodeFnBackward" and float((zd - zg).abs().max()) < 1e-5


def test_train_driver_with_unfrozen_encoder(dev, tmp_path, monkeypatch):
    """`train.py --usepretrainedvae ""` (the reference's flag is an untyped string: only the empty value is false): the
    encoder's weights move, the checkpoint carries them, the loss is finite."""
    import train as drv
    monkeypatch.chdir(tmp_path)
    save = str(tmp_path / "results")
    argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100", "--batch_size", "12",
            "--epochs", "2", "--save_path", save, "--synthetic", "12", "--random_init", "--checkpoint_path", "",
            "--usepretrainedvae", "", "--seed", "4"]
    args = drv.get_args(argv)
    assert not args.usepretrainedvae
    losses = drv.train(args)
    assert len(losses) > 0 and np.isfinite(losses).all()
    ck = torch.load(os.path.join(args.save_path, "model_1.pth"), map_location="cpu")
    w0 = synth.make_vae_state_dict(4)["encoder._conv_1.weight"]
    assert float((ck["model"]["encoder._conv_1.weight"] - w0).abs().max()) > 0
    assert len(ck["optimizer"]["state"]) == 48 + 12
    # "False" is a non-empty string: frozen, exactly as the reference's argparse line behaves
    assert drv.get_args([a if a != "" or i == 0 or argv[i - 1] != "--usepretrainedvae" else "False" for i, a in enumerate(argv)]).usepretrainedvae


_FLIP_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, {repo!r})
from t2ms_amd import synth
from t2ms_amd.train import mse_loss
from model.denoiser.transformer import Transformer
dev = torch.device("cuda", 0)
out = {{}}
for dtype in ("f32", "bf16"):
    m = Transformer(); m.load_state_dict(synth.make_dit_state_dict(2025), strict=True); m = m.to(dev).train().set_train_dtype(dtype)
    B = 40
    x = synth.make_latents(3, B).to(dev); t = (torch.arange(B, device=dev) * 7) % 100
    text = synth.make_text_embeddings(5, B).to(dev); target = synth.make_latents(4, B).to(dev)
    for rep in range(2):                                   # two steps: the flip parity restarts at the top of every forward
        for p in m.parameters():
            p.grad = None
        pred = m(input=x, t=t, text_input=text)
        mse_loss(pred, target).backward()
    out[dtype + "_pred"] = pred.detach().cpu().numpy()
    for n, p in m.named_parameters():
        if p.grad is not None:
            out[dtype + "_" + n] = p.grad.detach().cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_tile_walk_direction_changes_no_bit(dev, tmp_path):
    """consecutive launches of the training step walk their tiles in OPPOSITE directions so that a consumer starts on what
    the Infinity Cache still holds (T2S_TILE_FLIP, default on).  "Results do not depend on the order": tiles are independent and
    every partial-sum slot keeps its slab index -- so the prediction and all 48 gradients must be the same BITS with the flip off,
    in fp32 and in bf16 (child processes: the switch is read once)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "flip_ab.py"
    script.write_text(_FLIP_SCRIPT.format(repo=repo))
    outs = []
    for flag in ("1", "0"):
        dst = str(tmp_path / f"out_{flag}.npz")
        subprocess.run([sys.executable, str(script), dst], check=True, env=dict(os.environ, T2S_TILE_FLIP=flag), cwd=repo, timeout=300)
        outs.append(np.load(dst))
    assert set(outs[0].files) == set(outs[1].files) and len(outs[0].files) == 2 * 49
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k
