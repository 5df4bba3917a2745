"""Series/s of the fused sampler at small batches in both arithmetics (100-step DDPM, L = 96): where the chip is not filled."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                    # noqa: E402
from t2ms_amd import synth                      # noqa: E402
from t2ms_amd.sampler import Sampler            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model, vae = bench.build_models(dev)
    out = {}
    for math in ("f32", "bf16x3"):
        for B in (1, 2, 4, 8, 16, 32):
            text = synth.make_text_embeddings(3, B).to(dev)
            s = Sampler(model, vae.decoder, "ddpm", 100, 9.0, B, 96, dev, seed=1, math=math)
            s.run(text)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                s.run(text)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 3
            out[f"{math}_B{B}"] = {"series_per_s": round(B / el * 100 / 1000, 2), "ms_per_step": round(el * 10, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
