"""Reconstruction metrics on the GPU: MSE and WAPE as evaluation.py:166-206 computes them
from the arrays infer.py writes.  `python -m t2ms_amd.metrics <generation dir>` prints both for a run
directory ({save_path}/generation/{backbone}_{denoiser}_{dataset}_{cfg}_{steps}/[run_k/]x_1.npy, x_t.npy).
MRR, DTW / ED and the TS2Vec C-FID stay with the reference's evaluation.py (third-party dtaidistance /
learned encoder)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import _lib as L


def mse_wape(ori, gen, device="cuda"):
    """(MSE, WAPE, per_sample (N,2)) for (N, L, n_series) arrays (or tensors) of equal shape."""
    a = torch.as_tensor(np.asarray(ori) if not torch.is_tensor(ori) else ori).float()
    b = torch.as_tensor(np.asarray(gen) if not torch.is_tensor(gen) else gen).float()
    if a.shape != b.shape or a.dim() < 2:
        raise L.T2SError(f"mse_wape: shapes {tuple(a.shape)} vs {tuple(b.shape)}")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.T2SError("mse_wape: the metrics kernels run on a GPU; there is no CPU fallback")
    n = a.shape[0]
    a = a.reshape(n, -1).contiguous().to(dev)
    b = b.reshape(n, -1).contiguous().to(dev)
    per = torch.empty(n, 2, device=dev)
    out = torch.empty(2, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().t2s_eval_mse_wape(a.data_ptr(), b.data_ptr(), per.data_ptr(), out.data_ptr(), n, a.shape[1],
                                          L.stream_ptr(dev)), "t2s_eval_mse_wape")
    o = out.cpu()
    return float(o[0]), float(o[1]), per.cpu()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        sys.exit("usage: python -m t2ms_amd.metrics <directory holding x_1.npy and x_t.npy>")
    d = argv[0]
    ori, gen = np.load(os.path.join(d, "x_1.npy")), np.load(os.path.join(d, "x_t.npy"))
    mse, wape, _ = mse_wape(ori, gen)
    print(f"samples {ori.shape[0]}  MSE {mse:.6f}  WAPE {wape:.6f}")


if __name__ == "__main__":
    main()
