#!/usr/bin/env python3
"""Small-batch probe: series/s of the headline workload (1000-step CFG DDPM + decode) at the per-GPU shard sizes of a
STRONG-scaling run (256 series split over 1/2/4/8 GPUs = 256/128/64/32 per GPU) plus the in-situ per-kernel times of
one CFG forward at each size.  Sampling has no collective, so the 8-GPU strong curve is decided by these numbers.
    python tools/strong_probe.py [--batches 256,128,64,32] [--diffusion-steps 1000] [--lanes 0]
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,128,64,32")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, vae = bench.build_models(dev)
    out = {}
    for B in [int(b) for b in args.batches.split(",")]:
        s = Sampler(model, vae.decoder, "ddpm", args.diffusion_steps, 9.0, B, 96, dev, use_graph=True, seed=2025, row0=0,
                    lanes=args.lanes)
        text = synth.make_text_embeddings(2025, B).to(dev)
        s.run(text, decode=True)
        torch.cuda.synchronize(dev)
        best = 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            s.run_inplace(decode=True)
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        kt = bench.time_kernels_in_situ(model, dev, torch.randn(B, 64, 30, device=dev), text)
        out[B] = {"series_per_s": B / best, "ms_per_cfg_step": best / args.diffusion_steps * 1e3, **kt}
        print(B, json.dumps(out[B]), flush=True)
        del s
    ref = out.get(256)
    if ref:
        for B, r in out.items():
            print(f"B={B}: {r['series_per_s']:.2f} series/s, strong efficiency vs B=256 = "
                  f"{r['series_per_s'] / ref['series_per_s']:.3f}")


if __name__ == "__main__":
    main()
