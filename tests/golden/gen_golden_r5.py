#!/usr/bin/env python3
"""Round-5 fixtures: FILES THE REFERENCE WROTE, made by running the reference in the build container
(rules as in gen_golden.py: the reference never travels; what is committed are tensors + class paths the reference pickled,
and arrays of its outputs).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_r5.py

Writes under tests/golden/ref_made/
  final_model.pth    the whole-module LA-VAE pickle exactly as pretrained_lavae_unified.py:158 writes it --
                     `torch.save(model, .../final_model.pth)` of the reference's `model.pretrained.vqvae.vqvae` -- at SMALL
                     hyper-parameters (block_hidden_size 16, res_hidden_size 32, num_residual_layers 2, embedding_dim 64:
                     54 KB), torch-default init under a fixed seed.  infer.py:39 / train.py:22 load this with
                     weights_only=False.
  model_0.pth.gz     the checkpoint the reference's OWN train.train(args) wrote (train.py:134-136) after one epoch = one batch
                     = one AdamW step on the CPU: dict(model = Transformer.state_dict() incl. the grafted encoder.*, optimizer =
                     AdamW.state_dict(), epoch = 0, loss_list = [loss]).  gzip only because torch.save does not compress
                     (the test gunzips it into a temp dir: the bytes the reference wrote).  To keep the file small the
                     generator wraps torch.nn.init.xavier_uniform_ so the DiT's initial weights land on a 1/256 grid
                     (values are irrelevant to what this fixture pins: keys, dtypes, shapes, nesting, optimizer layout, resume
                     semantics); adaLN-Zero init makes the first step's gradients -- and so the Adam moments -- exactly zero
                     for the attention / MLP weights, which also compresses.
and tests/golden/ref_made.npz
  LA-VAE: the reference's encoder / decoder outputs of that module for L in {24, 96}, B = 3.
  checkpoint: loss_list, epoch, the optimizer's param_groups scalars and per-index step counts.
  resume: train.py:42-47 replayed with the reference classes (fresh model + AdamW + OneCycleLR, load_state_dict of both), then
          ONE step of train.py:118-127 on recorded inputs (x_t, t, emb, noise_gt): the loss, every gradient's norm, three whole
          gradient tensors (an adaLN bias, fc2 of the last block, patch_emb), lr, and a strided sample of every parameter after optimizer.step().
"""
import gzip
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "ref_made")
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

SMALL = dict(block_hidden_size=16, num_residual_layers=2, res_hidden_size=32, embedding_dim=64)


def main():
    torch.set_num_threads(8)
    from gen_golden import _install_timm_stub
    from t2ms_amd import synth
    _install_timm_stub()
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    for k in [k for k in sys.modules if k.split(".")[0] in ("model", "datafactory", "evaluate", "evaluation", "train", "infer")]:
        del sys.modules[k]
    work = tempfile.mkdtemp(prefix="t2s_refmade_")
    os.chdir(work)                      # '' on sys.path must not resolve to the repo root
    sys.path.insert(0, REF)
    from model.pretrained.vqvae import vqvae
    import train as ref_train
    for mod in ("model.pretrained.vqvae", "model.denoiser.transformer", "model.backbone.DDPM", "datafactory.dataloader", "train"):
        assert sys.modules[mod].__file__.startswith(REF + os.sep), (mod, sys.modules[mod].__file__)
    os.makedirs(OUT, exist_ok=True)
    out = {}

    # ---------------------------------------------------------------- the LA-VAE pickle (pretrained_lavae_unified.py:158)
    torch.manual_seed(2025)
    vae = vqvae(types.SimpleNamespace(**SMALL)).eval()
    vae_dir = os.path.join(work, "results", "saved_pretrained_models", "datasetETTh1_epoch2000")
    os.makedirs(vae_dir)
    torch.save(vae, os.path.join(vae_dir, "final_model.pth"))
    shutil.copy(os.path.join(vae_dir, "final_model.pth"), os.path.join(OUT, "final_model.pth"))
    with torch.no_grad():
        for L in (24, 96):
            xs = synth.make_series(500 + L, 3, L)
            z, before = vae.encoder(xs)
            rec, after = vae.decoder(z, length=L)
            zr = synth.make_latents(600 + L, 3)
            rec_r, after_r = vae.decoder(zr, length=L)
            out.update({f"vae_z_{L}": z, f"vae_before_{L}": before, f"vae_rec_{L}": rec, f"vae_after_{L}": after,
                        f"vae_rec_rand_{L}": rec_r, f"vae_after_rand_{L}": after_r})

    # ---------------------------------------------------------------- the checkpoint: the reference's own train() on the CPU
    os.makedirs(os.path.join(work, "Data", "our"))
    shutil.copy(os.path.join(HERE, "dataset_csv", "embedding_cleaned_ETTh1_24.csv"), os.path.join(work, "Data", "our"))
    real_xavier = torch.nn.init.xavier_uniform_

    def xavier_on_a_grid(tensor, gain=1.0, generator=None):
        real_xavier(tensor, gain=gain)
        with torch.no_grad():
            tensor.mul_(256.0).round_().div_(256.0)
        return tensor
    torch.nn.init.xavier_uniform_ = xavier_on_a_grid
    save_path = os.path.join(work, "results", "denoiser_results", "checkpoints", "ddpm_DiT_ETTh1_24")
    args = types.SimpleNamespace(checkpoint_path=None, dataset_name="ETTh1_24", batch_size=60, epochs=1, save_path=save_path,
                                 mix_train=False, usepretrainedvae=True, total_step=100, backbone="ddpm", denoiser="DiT",
                                 device="cpu", pretrained_model_path=os.path.join(vae_dir, "final_model.pth"))
    torch.manual_seed(7)
    np.random.seed(7)
    try:
        ref_train.train(args)
    finally:
        torch.nn.init.xavier_uniform_ = real_xavier
    ck_path = os.path.join(save_path, "model_0.pth")
    assert os.path.exists(ck_path)
    with open(ck_path, "rb") as f, gzip.GzipFile(os.path.join(OUT, "model_0.pth.gz"), "wb", compresslevel=9, mtime=0) as g:
        shutil.copyfileobj(f, g)
    ck = torch.load(ck_path, map_location="cpu")
    out["ck_loss_list"] = np.asarray(ck["loss_list"], dtype=np.float64)
    out["ck_epoch"] = np.asarray(ck["epoch"])
    out["ck_model_keys"] = np.asarray(list(ck["model"].keys()))
    pg = ck["optimizer"]["param_groups"][0]
    out["ck_pg_keys"] = np.asarray(sorted(pg.keys()))
    out["ck_pg_lr"] = np.asarray([pg["lr"], pg["initial_lr"], pg["max_lr"], pg["min_lr"], pg["weight_decay"], pg["eps"]], dtype=np.float64)
    out["ck_pg_betas"] = np.asarray(pg["betas"], dtype=np.float64)
    out["ck_pg_params"] = np.asarray(pg["params"])
    st = ck["optimizer"]["state"]
    out["ck_state_idx"] = np.asarray(sorted(st.keys()))
    out["ck_state_steps"] = np.asarray([float(st[i]["step"]) for i in sorted(st.keys())])

    # ---------------------------------------------------------------- resume (train.py:16-47) + one recorded step (:118-127)
    from model.denoiser.transformer import Transformer
    from model.backbone.DDPM import DDPM
    from torch.optim import AdamW, lr_scheduler
    torch.manual_seed(11)
    model = Transformer()
    pretrained = torch.load(os.path.join(vae_dir, "final_model.pth"), map_location="cpu", weights_only=False)
    pretrained.float()
    model.encoder = pretrained.encoder
    for name, p in model.named_parameters():
        if "encoder" in name:
            p.requires_grad = False
    optimizer = AdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
    scheduler = lr_scheduler.OneCycleLR(optimizer, max_lr=1e-4, total_steps=1 * 2)            # len(dataloader) * epochs, epochs = 2
    model.load_state_dict(ck["model"])
    optimizer.load_state_dict(ck["optimizer"])
    backbone = DDPM(100, "cpu")
    B = 4
    rs = np.random.RandomState(321)
    x_1 = synth.make_latents(77, B)
    t = torch.from_numpy(rs.randint(0, 100, size=B)).long()
    noise_gt = torch.from_numpy(rs.randn(B, 64, 30).astype(np.float32))
    emb = synth.make_text_embeddings(78, B)
    x_t, _ = backbone.q_sample(x_1, t, noise_gt)
    optimizer.zero_grad()
    pred = model(input=x_t, t=t, text_input=emb)
    loss = backbone.loss(pred, noise_gt)
    loss.backward()
    out["rs_lr"] = np.asarray(optimizer.param_groups[0]["lr"], dtype=np.float64)
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    out["rs_grad_names"] = np.asarray(names)
    grads = dict((n, p.grad.detach().clone()) for n, p in model.named_parameters() if p.grad is not None)
    out["rs_grad_norms"] = np.asarray([float(grads[n].double().norm()) for n in names])
    out["rs_grad_maxabs"] = np.asarray([float(grads[n].abs().max()) for n in names])
    for n in ("layers.0.adaLN_modulation.1.bias", "layers.3.mlp.fc2.weight", "patch_emb.weight"):
        out["rs_grad__" + n.replace(".", "__")] = grads[n]
    optimizer.step()
    scheduler.step()
    out.update(rs_x_t=x_t, rs_t=t, rs_noise=noise_gt, rs_emb=emb, rs_pred_sample=pred.detach().flatten()[::53].clone(),
               rs_loss=np.asarray(float(loss), dtype=np.float64))
    for n, p in model.named_parameters():
        if p.grad is not None:
            out["rs_after__" + n.replace(".", "__")] = p.detach().flatten()[::97].clone()
    st2 = optimizer.state_dict()["state"]
    out["rs_state_steps"] = np.asarray([float(st2[i]["step"]) for i in sorted(st2.keys())])

    arrs = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "ref_made.npz"), **arrs)
    for f in ("final_model.pth", "model_0.pth.gz"):
        print(f"ref_made/{f}  {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")
    print(f"ref_made.npz  {os.path.getsize(os.path.join(HERE, 'ref_made.npz')) / 1024:.1f} KiB")
    print("loss_list:", ck["loss_list"], " resumed-step loss:", float(loss), " lr:", float(out["rs_lr"]))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
