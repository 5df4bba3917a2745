#!/usr/bin/env python3
"""The bf16 training step across per-GPU batch sizes: BASELINE configs[3] is 1152 rows per GPU at fixed L, the reference's
default mix-train (train.py:145, dataloader.py:80-99) makes three length groups of ~3072 rows per 9216-row batch -- 3072
rows per step on one GPU, 384 per GPU on eight.  Prints ms / step, us / row and the per-class kernel times per size.

    python tools/train_shapes.py [--batches 192,384,768,1152,2304,3072] [--steps 12]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="192,384,768,1152,1536,2304,3072")
    ap.add_argument("--steps", type=int, default=12)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rows = []
    for b in [int(x) for x in a.batches.split(",")]:
        o = bench.train_leg(dev, None, 0, 1, batch=b, steps=a.steps, warmup=3, mix_shard=False)
        kc = {k: round(v["ms_per_step"], 3) for k, v in o["kernel_classes"].items()}
        rows.append({"batch": b, "ms_per_step": round(o["ms_per_step"], 3), "us_per_row": round(o["ms_per_step"] * 1e3 / b, 3),
                     "samples_per_s": round(o["value"]), "kernel_ms": round(sum(kc.values()), 3), "classes": kc})
        print(json.dumps(rows[-1]), flush=True)
        torch.cuda.empty_cache()
    print(json.dumps({"train_shapes": rows}))


if __name__ == "__main__":
    main()
