#!/usr/bin/env python3
"""Two host threads driving two samplers (own models / handles) on one device, many fresh-sampler + replay rounds: every
result must equal what the sampler gives alone, and no call may fail (tests/test_hip_parity.py holds a short cut of this).
    python tools/stress_threads.py [--reps 150]
"""
import argparse
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import synth  # noqa: E402
from t2ms_amd.sampler import Sampler  # noqa: E402
from model.denoiser.transformer import Transformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=150)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _, vae = bench.build_models(dev)
    models = []
    for seed in (31337, 4242):
        m = Transformer()
        m.load_state_dict(synth.make_dit_state_dict(seed, gain=0.7), strict=True)
        models.append(m.to(dev).eval())
    texts = [synth.make_text_embeddings(21 + i, 64).to(dev) for i in range(2)]
    want = []
    for i in range(2):
        s = Sampler(models[i], vae.decoder, "ddpm", 5, 9.0, 64, 48, dev, seed=100 + i, lanes=2)
        lat, ser, _ = s.run(texts[i])
        want.append((lat.clone(), ser.clone()))
    torch.cuda.synchronize(dev)
    bad, errors = [0, 0], []

    def work(i):
        try:
            torch.cuda.set_device(dev)
            for rep in range(a.reps):
                s = Sampler(models[i], vae.decoder, "ddpm", 5, 9.0, 64, 48, dev, seed=100 + i, lanes=2 if rep % 3 else 1)
                for _ in range(2):
                    lat, ser, _ = s.run(texts[i])
                    if not (torch.equal(lat, want[i][0]) and torch.equal(ser, want[i][1])):
                        bad[i] += 1
            torch.cuda.synchronize(dev)
        except Exception as e:                       # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    print("mismatches", bad, "errors", errors)
    sys.exit(1 if (sum(bad) or errors) else 0)


if __name__ == "__main__":
    main()
