#!/usr/bin/env python3
"""The training leg of bench.py on its own (BASELINE configs[3] shape: DiT train step of train.py, bf16, B=1152/GPU,
L=96, cached latents) -- the command the rocprofv3 passes of tools/collect_profiles.sh wrap.

    python tools/bench_train.py --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py ...
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from t2ms_amd import dist as tdist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1152)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16")
    a = ap.parse_args()
    rank, _, world = tdist.env_world()
    torch.cuda.set_device(tdist.local_device_index())
    dev = torch.device("cuda", tdist.local_device_index())
    dist = tdist.init("nccl", dev)
    out = bench.train_leg(dev, dist, rank, world, batch=a.batch, steps=a.steps, warmup=a.warmup, dtype=a.dtype,
                          mix_shard=False)     # the PMC passes count bytes per step: only whole steps of ONE shape
    if rank == 0:
        print(json.dumps(out))
    tdist.barrier(dist, dev)


if __name__ == "__main__":
    main()
