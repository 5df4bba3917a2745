"""Evaluation metrics on the GPU, computed from the arrays infer.py writes exactly as the reference's
evaluation.py defines them: MSE / WAPE (:166-206), ED (:137-150), DTW (:152-163), MRR (:21-45) and CRPS (:51-83) over
the run_0..run_k repetitions `--run_multi` writes, and the TS2Vec encoder forward + FID of the C-FID metric
(:127-135,238-243; evaluate/ts2vec.py:352-399).  `python -m t2ms_amd.metrics <generation dir>` prints them for a
directory holding x_1.npy and x_t.npy ({save_path}/generation/{backbone}_{denoiser}_{dataset}_{cfg}_{steps}/[run_k/])
and the run_* sub-directories.

C-FID end to end: evaluation.py TRAINS the TS2Vec encoder at evaluation time (initialize_ts2vec, 200 contrastive
iterations from a random initialisation, :238); `t2ms_amd.ts2vec` restates that training (torch autograd plumbing, pinned
to the reference's loss curve) and `TS2VecEncoder` here runs the forward of the trained encoder on the HIP kernel, so the
CLI prints C-FID from the .npy files alone.  (Values are comparable only for the same trained encoder / seeds.)"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

import ctypes as C

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


def _pair(ori, gen, what):
    a = torch.as_tensor(np.asarray(ori) if not torch.is_tensor(ori) else ori).float()
    b = torch.as_tensor(np.asarray(gen) if not torch.is_tensor(gen) else gen).float()
    if a.shape != b.shape or a.dim() != 3:
        raise L.T2SError(f"{what}: expected two (N, L, n_series) arrays, got {tuple(a.shape)} and {tuple(b.shape)}")
    return a, b


def _gpu(device, what):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise L.T2SError(f"{what}: the metrics kernels run on a GPU; there is no CPU fallback")
    return dev


def ed(ori, gen, device="cuda"):
    """(ED, per_sample (N,)): calculate_ed, evaluation.py:137-150."""
    a, b = _pair(ori, gen, "ed")
    dev = _gpu(device, "ed")
    a, b = a.contiguous().to(dev), b.contiguous().to(dev)
    per, out = torch.empty(a.shape[0], device=dev), torch.empty(1, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().t2s_eval_ed(a.data_ptr(), b.data_ptr(), per.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1],
                                    a.shape[2], L.stream_ptr(dev)), "t2s_eval_ed")
    return float(out.cpu()[0]), per.cpu()


def dtw(ori, gen, device="cuda"):
    """(DTW, per_sample (N,)): calculate_dtw, evaluation.py:152-163 (dtaidistance dtw_ndim.distance)."""
    a, b = _pair(ori, gen, "dtw")
    dev = _gpu(device, "dtw")
    a, b = a.contiguous().to(dev), b.contiguous().to(dev)
    per, out = torch.empty(a.shape[0], device=dev), torch.empty(1, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().t2s_eval_dtw(a.data_ptr(), b.data_ptr(), per.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1],
                                     a.shape[2], L.stream_ptr(dev)), "t2s_eval_dtw")
    return float(out.cpu()[0]), per.cpu()


def _runs_first(ori, gens, what):
    a = torch.as_tensor(np.asarray(ori) if not torch.is_tensor(ori) else ori).float()
    if torch.is_tensor(gens) or isinstance(gens, np.ndarray):
        g = torch.as_tensor(np.asarray(gens) if not torch.is_tensor(gens) else gens).float()
        if g.dim() != a.dim() + 1 or g.shape[:-1] != a.shape:
            raise L.T2SError(f"{what}: generations {tuple(g.shape)} vs original {tuple(a.shape)}")
        g = g.movedim(-1, 0)
    else:
        g = torch.stack([torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).float() for x in gens])
        if g.shape[1:] != a.shape:
            raise L.T2SError(f"{what}: generations {tuple(g.shape[1:])} vs original {tuple(a.shape)}")
    return a, g


def crps(ori, gens, device="cuda"):
    """(CRPS, per_sample (N,)): calculate_crps, evaluation.py:51-83; ori (N, L, n_series), `gens` the G generated arrays
    (a sequence, run_0 first, or one (N, L, n_series, G) array as evaluation.py:311-313 stacks them)."""
    a, g = _runs_first(ori, gens, "crps")
    if a.dim() != 3:
        raise L.T2SError(f"crps: ori must be (N, L, n_series), got {tuple(a.shape)}")
    dev = _gpu(device, "crps")
    a, g = a.contiguous().to(dev), g.contiguous().to(dev)
    per, out = torch.empty(a.shape[0], device=dev), torch.empty(1, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().t2s_eval_crps(a.data_ptr(), g.data_ptr(), per.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1],
                                      a.shape[2], g.shape[0], L.stream_ptr(dev)), "t2s_eval_crps")
    return float(out.cpu()[0]), per.cpu()


class TS2VecEncoder:
    """Forward of evaluate/ts2vec.py's TSEncoder (:352-399, eval mode, mask 'all_true') on the GPU from its state dict
    (keys input_fc.*, feature_extractor.net.<i>.conv{1,2}.conv.*, feature_extractor.net.<depth>.projector.*)."""

    def __init__(self, state_dict, device="cuda"):
        self.device = _gpu(device, "TS2VecEncoder")
        sd = {k: torch.as_tensor(v).float().contiguous().to(self.device) for k, v in state_dict.items()}
        depth = 0
        while f"feature_extractor.net.{depth + 1}.conv1.conv.weight" in sd:
            depth += 1
        w = L.Ts2vecWeights()
        w.hidden, w.input_dims = sd["input_fc.weight"].shape
        w.depth = depth
        w.output_dims = sd[f"feature_extractor.net.{depth}.conv1.conv.weight"].shape[0]
        if depth >= L.TS2VEC_MAX_BLOCKS:
            raise L.T2SError(f"TS2VecEncoder: depth {depth} exceeds {L.TS2VEC_MAX_BLOCKS - 1}")
        w.fc_w, w.fc_b = sd["input_fc.weight"].data_ptr(), sd["input_fc.bias"].data_ptr()
        for i in range(depth + 1):
            p = f"feature_extractor.net.{i}."
            if f"{p}projector.weight" in sd and i != depth:
                raise L.T2SError("TS2VecEncoder: a projector inside the stack (unequal hidden widths) is not supported")
            w.conv1_w[i], w.conv1_b[i] = sd[p + "conv1.conv.weight"].data_ptr(), sd[p + "conv1.conv.bias"].data_ptr()
            w.conv2_w[i], w.conv2_b[i] = sd[p + "conv2.conv.weight"].data_ptr(), sd[p + "conv2.conv.bias"].data_ptr()
        p = f"feature_extractor.net.{depth}.projector."
        w.proj_w, w.proj_b = sd[p + "weight"].data_ptr(), sd[p + "bias"].data_ptr()
        self._w, self._keep = w, sd

    def encode(self, x, encoding_window="full_series"):
        """x (B, T, input_dims) -> (B, output_dims) for 'full_series' (TS2Vec.encode, ts2vec.py:236-245), or the
        per-step representations (B, T, output_dims) for encoding_window=None."""
        if encoding_window not in ("full_series", None):
            raise L.T2SError("TS2VecEncoder.encode: encoding_window must be 'full_series' or None")
        xs = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).float().contiguous().to(self.device)
        if xs.dim() != 3 or xs.shape[2] != self._w.input_dims:
            raise L.T2SError(f"TS2VecEncoder.encode: x must be (B, T, {self._w.input_dims}), got {tuple(xs.shape)}")
        B, T = xs.shape[0], xs.shape[1]
        full = torch.empty(B, self._w.output_dims, device=self.device)
        rep = torch.empty(B, T, self._w.output_dims, device=self.device) if encoding_window is None else None
        with torch.cuda.device(self.device):
            L.check(L.lib().t2s_ts2vec_encode(C.byref(self._w), xs.data_ptr(), None if rep is None else rep.data_ptr(),
                                              full.data_ptr(), B, T, L.stream_ptr(self.device)), "t2s_ts2vec_encode")
        return full if rep is None else rep


def fid(act1, act2):
    """calculate_fid (evaluation.py:127-135) on two (N, C) activation arrays: host fp64 (a C x C matrix square root)."""
    from scipy.linalg import sqrtm
    a1 = np.asarray(act1.cpu() if torch.is_tensor(act1) else act1, dtype=np.float64)
    a2 = np.asarray(act2.cpu() if torch.is_tensor(act2) else act2, dtype=np.float64)
    mu1, s1 = a1.mean(axis=0), np.cov(a1, rowvar=False)
    mu2, s2 = a2.mean(axis=0), np.cov(a2, rowvar=False)
    covmean = sqrtm(s1.dot(s2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum((mu1 - mu2) ** 2.0) + np.trace(s1 + s2 - 2.0 * covmean))


def cfid(ori, gen, encoder: "TS2VecEncoder"):
    """C-FID as evaluation.py:238-243 computes it once the encoder exists: FID of the full-series representations of the
    original and the generated (N, L, n_series) arrays."""
    return fid(encoder.encode(ori), encoder.encode(gen))


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
        print(f"samples {ori.shape[0]}  MSE {mse:.6f}  WAPE {wape:.6f}  ED {ed(ori, gen)[0]:.6f}  DTW {dtw(ori, gen)[0]:.6f}")
        if ori.shape[0] >= 8 and os.environ.get("T2S_METRICS_CFID", "1") not in ("", "0"):
            # evaluation.py:238-243: train TS2Vec on the original series (200 contrastive iterations), encode both sets
            # with it, FID of the representations.  The encoder is trained per call from a random initialisation, as in
            # the reference, so the value varies from run to run unless torch / numpy are seeded by the caller.
            from .ts2vec import initialize_ts2vec
            o3 = ori if ori.ndim == 3 else ori[:, :, None]
            g3 = gen if gen.ndim == 3 else gen[:, :, None]
            model = initialize_ts2vec(o3.astype(np.float32), device="cuda")
            print(f"samples {ori.shape[0]}  C-FID {fid(model.encode(o3, encoding_window='full_series'), model.encode(g3, encoding_window='full_series')):.6f}"
                  f"  (TS2Vec trained {model.n_iters} iterations on the original series)")
    runs = sorted((r for r in os.listdir(d) if r.startswith("run_") and r[4:].isdigit()), key=lambda r: int(r[4:]))
    if runs:
        ori = np.load(os.path.join(d, runs[-1], "x_1.npy"))                                  # evaluation.py:304-314
        gens = [np.load(os.path.join(d, r, "x_t.npy")) for r in runs]
        m, _, _ = mrr(ori, gens)
        print(f"samples {ori.shape[0]}  runs {len(runs)}  MRR {m:.6f}  CRPS {crps(ori, gens)[0]:.6f}")


if __name__ == "__main__":
    main()
