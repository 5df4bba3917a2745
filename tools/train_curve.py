#!/usr/bin/env python3
"""Loss curves of the DiT training step in fp32 and bf16 from the same initial weights, data order and noise stream
(train.train_step on synthetic rows, B per step = --batch, DDPM T=100, AdamW 1e-4): the bf16 path has to track the fp32 one.

    python tools/train_curve.py --steps 300 --batch 1152 > profiles/<tag>_train_curve.json
"""
import argparse
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import train as drv  # noqa: E402
from model.backbone.DDPM import DDPM  # noqa: E402
from t2ms_amd import latent_cache, synth  # noqa: E402
from t2ms_amd.train import T2SAdamW  # noqa: E402


def run(dtype, steps, batch, n_rows, dev, lr):
    model, vae = bench.build_models(dev)
    model.train().set_train_dtype(dtype)
    model.encoder = vae.encoder
    for n, p in model.named_parameters():
        if "encoder" in n:
            p.requires_grad = False
    opt = T2SAdamW(model.parameters(), lr=lr, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    args = types.SimpleNamespace(backbone="ddpm", total_step=100, seed=2025)
    x = synth.make_series(7, n_rows, 96)
    text = synth.make_text_embeddings(7, n_rows)
    lat = latent_cache.encode_all(model.encoder, x, dev)
    torch.manual_seed(2025)
    gen = torch.Generator().manual_seed(99)
    losses = []
    for s in range(steps):
        idx = torch.randint(0, n_rows, (batch,), generator=gen)
        loss = drv.train_step(model, ddpm, opt, None, args, x[idx], text[idx], dev, 0, 1, lat, idx, s)
        losses.append(loss)
    return [float(l.detach()) for l in losses]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=1152)
    ap.add_argument("--rows", type=int, default=4608)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {"steps": a.steps, "batch": a.batch, "rows": a.rows, "lr": a.lr,
           "note": "same initial weights (synth seed), same batches, same t / CFG-coin / Philox noise streams; only the arithmetic differs"}
    for dt in ("f32", "bf16"):
        ls = run(dt, a.steps, a.batch, a.rows, dev, a.lr)
        out[dt] = {"every_10th": [round(v, 5) for v in ls[::10]], "last_20_mean": sum(ls[-20:]) / 20, "first": ls[0]}
    f, b = out["f32"], out["bf16"]
    out["max_rel_gap_every_10th"] = max(abs(x - y) / x for x, y in zip(f["every_10th"], b["every_10th"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
