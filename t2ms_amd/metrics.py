"""Evaluation metrics on the GPU: MSE and WAPE as evaluation.py:166-206 computes them from
the arrays infer.py writes, and MRR (evaluation.py:21-45) over the run_0..run_k repetitions `--run_multi`
writes.  `python -m t2ms_amd.metrics <generation dir>` prints MSE / WAPE for a directory holding x_1.npy and
x_t.npy ({save_path}/generation/{backbone}_{denoiser}_{dataset}_{cfg}_{steps}/[run_k/]) and MRR when it holds
run_* sub-directories.  DTW / ED and the TS2Vec C-FID stay with the reference's evaluation.py (third-party
dtaidistance / an encoder trained at evaluation time)."""
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


def mrr(ori, gens, threshold=0.5, device="cuda"):
    """(MRR, sims (N,G), score (N,)) for ori (N, L, n_series) and the G generated arrays `gens` (a sequence of
    arrays of ori's shape, run_0 first, or one (N, L, n_series, G) array as evaluation.py:311-313 stacks them)."""
    a = torch.as_tensor(np.asarray(ori) if not torch.is_tensor(ori) else ori).float()
    if torch.is_tensor(gens) or isinstance(gens, np.ndarray):
        g = torch.as_tensor(np.asarray(gens) if not torch.is_tensor(gens) else gens).float()
        if g.dim() != a.dim() + 1 or g.shape[:-1] != a.shape:
            raise L.T2SError(f"mrr: generations {tuple(g.shape)} vs original {tuple(a.shape)}")
        g = g.movedim(-1, 0)
    else:
        g = torch.stack([torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).float() for x in gens])
        if g.shape[1:] != a.shape:
            raise L.T2SError(f"mrr: generations {tuple(g.shape[1:])} vs original {tuple(a.shape)}")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.T2SError("mrr: the metrics kernels run on a GPU; there is no CPU fallback")
    n, runs = a.shape[0], g.shape[0]
    a = a.reshape(n, -1).contiguous().to(dev)
    g = g.reshape(runs, n, -1).contiguous().to(dev)
    sims = torch.empty(n, runs, device=dev)
    score = torch.empty(n, device=dev)
    out = torch.empty(1, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().t2s_eval_mrr(a.data_ptr(), g.data_ptr(), sims.data_ptr(), score.data_ptr(), out.data_ptr(),
                                     n, a.shape[1], runs, float(threshold), L.stream_ptr(dev)), "t2s_eval_mrr")
    return float(out.cpu()[0]), sims.cpu(), score.cpu()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        sys.exit("usage: python -m t2ms_amd.metrics <directory holding x_1.npy and x_t.npy and / or run_*/>")
    d = argv[0]
    if os.path.exists(os.path.join(d, "x_t.npy")):
        x1 = os.path.join(d, "x_1.npy")
        ori = np.load(x1 if os.path.exists(x1) else os.path.join(d, "run_0", "x_1.npy"))   # evaluation.py:285-286
        gen = np.load(os.path.join(d, "x_t.npy"))
        mse, wape, _ = mse_wape(ori, gen)
        print(f"samples {ori.shape[0]}  MSE {mse:.6f}  WAPE {wape:.6f}")
    runs = sorted((r for r in os.listdir(d) if r.startswith("run_") and r[4:].isdigit()), key=lambda r: int(r[4:]))
    if runs:
        ori = np.load(os.path.join(d, runs[-1], "x_1.npy"))                                  # evaluation.py:304-314
        m, _, _ = mrr(ori, [np.load(os.path.join(d, r, "x_t.npy")) for r in runs])
        print(f"samples {ori.shape[0]}  runs {len(runs)}  MRR {m:.6f}")


if __name__ == "__main__":
    main()
