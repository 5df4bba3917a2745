#!/usr/bin/env python3
"""Where a training step's wall time goes, phase by phase (device drained after each phase), at several batch sizes."""
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import latent_cache, synth  # noqa: E402
from t2ms_amd.sampler import philox_normal  # noqa: E402
from t2ms_amd.train import T2SAdamW  # noqa: E402
from model.backbone.DDPM import DDPM  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2304,3072").split(",")]:
        model, vae = bench.build_models(dev)
        model.train().set_train_dtype("bf16")
        model.encoder = vae.encoder
        for n, p in model.named_parameters():
            if "encoder" in n:
                p.requires_grad = False
        opt = T2SAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
        ddpm = DDPM(100, dev)
        lat = latent_cache.encode_all(model.encoder, synth.make_series(1, B, 96), dev)
        emb = synth.make_text_embeddings(1, B).to(dev)
        idx = torch.arange(B, device=dev)
        acc = {}

        def phase(name, fn, sync):
            t0 = time.perf_counter()
            r = fn()
            if sync:
                torch.cuda.synchronize(dev)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r

        for sync in (True, False):
            acc.clear()
            n_steps = 6
            for step in range(n_steps + 2):
                if step == 2:
                    torch.cuda.synchronize(dev)
                    acc.clear()
                    t_all = time.perf_counter()
                opt.zero_grad()
                z = phase("gather", lambda: lat[idx], sync)
                noise = phase("philox", lambda: philox_normal(B, 1920, 7, step, 0, dev).view_as(z), sync)
                t = torch.randint(0, 100, (B,), device=dev)
                x_t = phase("q_sample", lambda: ddpm.q_sample(z, t, noise)[0], sync)
                pred = phase("forward", lambda: model(input=x_t, t=t, text_input=emb), sync)
                loss = phase("loss", lambda: ddpm.loss(pred, noise), sync)
                phase("backward", lambda: loss.backward(), sync)
                phase("adamw", lambda: opt.step(), sync)
            torch.cuda.synchronize(dev)
            total = (time.perf_counter() - t_all) / n_steps
            print(json.dumps({"batch": B, "sync_after_each_phase": sync, "ms_per_step": round(total * 1e3, 3),
                              "phases_ms": {k: round(v / n_steps * 1e3, 3) for k, v in acc.items()}}), flush=True)
        del model, opt, lat
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
