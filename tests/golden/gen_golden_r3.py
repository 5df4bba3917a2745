#!/usr/bin/env python3
"""Round-3 fixtures, made by RUNNING THE REFERENCE in the build container (rules as in gen_golden.py / gen_golden_r2.py:
the reference never travels, only the arrays written here are committed).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_r3.py

Writes
  vae_long.npz   the reference LA-VAE (model/pretrained/vqvae.py:36-105, default hyper-parameters, seeded weights of
                 t2ms_amd.synth) on series LONGER than one LDS tile: L in {512, 2048} (the reference's SUSHI length,
                 dataloader.py:88-90 / evaluation.py:282), B in {1, 3}: encoder z (B,64,30) and `before`, decoder recon
                 and `after` -- `before` / `after` at every 7th / 11th position plus their fp64 per-row sums (the full
                 (B,64,L/4) arrays would be 0.4 MB each); and, for B = 1, a decode of a random latent.                     -> SURVEY 8a rows a16, a17
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def save(name, **arrs):
    arrs = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.set_num_threads(8)
    from t2ms_amd import synth
    # the repo's own `model/` package would shadow the reference's: take the repo off the path once synth is imported
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    for k in [k for k in sys.modules if k.split(".")[0] in ("model", "datafactory", "evaluate", "evaluation")]:
        del sys.modules[k]
    os.chdir(HERE)
    sys.path.insert(0, REF)
    from model.pretrained.vqvae import vqvae
    assert sys.modules["model.pretrained.vqvae"].__file__.startswith(REF + os.sep), sys.modules["model.pretrained.vqvae"].__file__
    ns = types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64)
    vae = vqvae(ns).eval()
    print("VAE load_state_dict(strict):", vae.load_state_dict(synth.make_vae_state_dict(2025), strict=True))
    out = {}
    with torch.no_grad():
        for L in (512, 2048):
            for B in (1, 3):
                xs = synth.make_series(100 + L + B, B, L)
                z, before = vae.encoder(xs)
                rec, after = vae.decoder(z, length=L)
                zr = synth.make_latents(300 + L, B)
                rec2, after2 = vae.decoder(zr, length=L)
                key = f"{L}_{B}"
                out[f"z_{key}"] = z
                out[f"rec_{key}"] = rec
                if B == 1:
                    out[f"rec_rand_{key}"] = rec2
                st = 7 if L == 512 else 11          # coprime with the tile cores (26 / 24 positions): every seam offset occurs
                for name, t in (("before", before), ("after", after)):
                    out[f"{name}_s{st}_{key}"] = t[:, :, ::st].contiguous()
                    out[f"{name}_rowsum_{key}"] = t.double().sum(dim=2)
    save("vae_long", **out)

    # TS2Vec.fit (evaluate/ts2vec.py:73-160) as evaluation.py:238 configures it, on the CPU: per-iteration losses of the
    # first 12 iterations under fixed seeds, the averaged encoder's full-series representations, and the FID they give
    from evaluate.ts2vec import TS2Vec
    assert sys.modules["evaluate.ts2vec"].__file__.startswith(REF + os.sep)
    rs = np.random.RandomState(77)
    tt = np.linspace(0, 1, 24)[None, :, None]
    ori = (np.sin(2 * np.pi * (rs.uniform(1, 3, (24, 1, 1)) * tt + rs.uniform(0, 1, (24, 1, 1)))) * rs.uniform(0.3, 1, (24, 1, 1))
           + 0.05 * rs.randn(24, 24, 1)).astype(np.float32)
    gen = (ori + 0.15 * rs.randn(24, 24, 1)).astype(np.float32)
    torch.manual_seed(7)
    np.random.seed(7)
    losses = []
    m = TS2Vec(input_dims=1, device="cpu", batch_size=8, lr=0.001, output_dims=100, max_train_length=3000,
               after_iter_callback=lambda model, loss: losses.append(loss))
    log = m.fit(ori.copy(), n_iters=12, verbose=False)
    r_ori = m.encode(ori.copy(), encoding_window="full_series")
    r_gen = m.encode(gen.copy(), encoding_window="full_series")
    # the default n_iters = 200 run evaluation.py performs: only its end state (loss of the last iterations)
    torch.manual_seed(8)
    np.random.seed(8)
    losses200 = []
    m2 = TS2Vec(input_dims=1, device="cpu", batch_size=8, lr=0.001, output_dims=100, max_train_length=3000,
                after_iter_callback=lambda model, loss: losses200.append(loss))
    m2.fit(ori.copy(), verbose=False)
    save("ts2vec_fit", ori=ori, gen=gen, losses=np.asarray(losses), epoch_log=np.asarray(log), repr_ori=r_ori, repr_gen=r_gen,
         losses200=np.asarray(losses200), n_iters200=np.asarray([m2.n_iters, m2.n_epochs]))


if __name__ == "__main__":
    main()
