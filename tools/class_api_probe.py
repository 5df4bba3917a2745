#!/usr/bin/env python3
"""ms per diffusion step of the reference-style class-API loop (infer.py:76-88 against the mirrors: two model(...) calls,
torch CFG glue, p_sample) next to the fused sampler, by batch size -- the reference's default loader batch is 2.
    python tools/class_api_probe.py [--batches 2,8,32,256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import synth  # noqa: E402
from t2ms_amd.sampler import Sampler  # noqa: E402
from model.backbone.DDPM import DDPM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="2,8,32,256")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    model, vae = bench.build_models(dev)
    T = 1000
    ddpm = DDPM(T, dev)
    for B in [int(b) for b in a.batches.split(",")]:
        emb = synth.make_text_embeddings(1, B).to(dev)
        x_t = torch.randn(B, 64, 30, device=dev)

        def steps(x_t, j0, n):
            for j in range(j0, j0 + n):
                t = torch.full((x_t.size(0),), T - 1 - j, dtype=torch.long, device=dev)
                u = model(input=x_t, t=t, text_input=None)
                c = model(input=x_t, t=t, text_input=emb)
                x_t = ddpm.p_sample(x_t, u + 9.0 * (c - u), t)
            return x_t

        with torch.no_grad():
            x_t = steps(x_t, 0, 20)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n = 200 if B <= 32 else 60
            x_t = steps(x_t, 20, n)
            host = (time.perf_counter() - t0) / n
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / n
        s = Sampler(model, vae.decoder, "ddpm", 200, 9.0, B, 96, dev, use_graph=True, seed=1)
        s.run(emb, decode=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        s.run_inplace(decode=False)
        torch.cuda.synchronize(dev)
        fused = (time.perf_counter() - t0) / 200
        print(json.dumps({"batch": B, "class_api_ms_per_step": round(wall * 1e3, 4), "host_enqueue_ms_per_step": round(host * 1e3, 4),
                          "fused_ms_per_step": round(fused * 1e3, 4), "ratio": round(wall / fused, 3)}), flush=True)


if __name__ == "__main__":
    main()
