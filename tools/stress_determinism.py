#!/usr/bin/env python3
"""Race screen for the hand-synchronised kernels (counted vmcnt / lgkmcnt, untracked LDS-DMA rings): every launch shape of the
DiT forward is repeated many times on the same inputs and must reproduce its first result BIT FOR BIT; interleaved with
other-shape launches and a second stream's traffic so that timing varies.  A DMA / barrier race shows up as a differing
tile sooner or later; a clean run is no proof, a differing run is a bug.
    python tools/stress_determinism.py [--rounds 40] [--math bf16x3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import synth  # noqa: E402
from t2ms_amd.sampler import Sampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--math", choices=["f32", "bf16x3"], default="f32", help="matrix arithmetic of the forward / the samplers")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model, vae = bench.build_models(dev)
    model.set_math(args.math)
    shapes = [1, 3, 7, 16, 25, 32, 50, 64, 100, 128, 256]
    inputs, first = {}, {}
    side = torch.cuda.Stream(dev)
    junk = torch.randn(64 << 20, device=dev)
    with torch.no_grad():
        for B in shapes:
            x = synth.make_latents(100 + B, B).to(dev)
            t = torch.full((B,), 500, dtype=torch.long, device=dev)
            text = synth.make_text_embeddings(100 + B, B).to(dev)
            inputs[B] = (x, t, text)
            first[B] = (model(input=x, t=t, text_input=text).clone(), model(input=x, t=t, text_input=None).clone())
        bad = 0
        for r in range(args.rounds):
            for B in shapes if r % 2 == 0 else reversed(shapes):
                x, t, text = inputs[B]
                with torch.cuda.stream(side):          # HBM / L2 noise from another queue
                    junk.mul_(1.0000001)
                if r % 4 < 2:           # conditional first: two plain forwards
                    c = model(input=x, t=t, text_input=text)
                    u = model(input=x, t=t, text_input=None)
                else:                   # the reference loop's order: the mirror runs the pair as ONE 2B-sequence CFG pass
                    u = model(input=x, t=t, text_input=None)
                    c = model(input=x, t=t, text_input=text)
                if not (torch.equal(c, first[B][0]) and torch.equal(u, first[B][1])):
                    bad += 1
                    print(f"MISMATCH round {r} B={B}: max diff {float((c - first[B][0]).abs().max()):.3e}", flush=True)
            if r % 10 == 0:
                print(f"round {r}: {bad} mismatches so far", flush=True)
        # the fused sampler (graph replay, lanes, table, 16- / 32-token tiles), short chains repeated
        for B in (8, 32, 64, 256):
            s = Sampler(model, vae.decoder, "ddpm", 25, 9.0, B, 96, dev, seed=5)
            text = synth.make_text_embeddings(7, B).to(dev)
            lat0, ser0, _ = s.run(text)
            for r in range(max(4, args.rounds // 4)):
                with torch.cuda.stream(side):
                    junk.mul_(1.0000001)
                lat, ser = s.run_inplace()
                torch.cuda.synchronize()
                if not (torch.equal(lat, lat0) and torch.equal(ser, ser0)):
                    bad += 1
                    print(f"MISMATCH sampler B={B} rep {r}: {float((lat - lat0).abs().max()):.3e}", flush=True)
            print(f"sampler B={B}: ok so far ({bad} mismatches)", flush=True)
    print("TOTAL MISMATCHES", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
