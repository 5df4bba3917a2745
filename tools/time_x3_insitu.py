"""Kernel breakdown (HIP events in situ) of a bf16x3 CFG forward at the headline shape; no result checks, so it also
serves the timing ablations of tools/exp_x3_ablate.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from t2ms_amd import synth

dev = torch.device("cuda:0")
model, _ = bench.build_models(dev)
model.set_math(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
B = 256
x = synth.make_latents(1, B).to(dev)
text = synth.make_text_embeddings(1, B).to(dev)
kt = bench.time_kernels_in_situ(model, dev, x, text, n_steps=6)
print("attention %.1f us  rows %.1f us  forward %.1f us" % (kt["attn_us"], kt["rows_us"], kt["forward_us"]))
