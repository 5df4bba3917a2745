"""T2S CSV dataset (reference datafactory/dataset.py:10-104), rewritten around numpy arrays.

Contract kept from the reference:
  * CSV columns `Text`, `OT` (a python-literal list per row), `TextEmbedding` (python list for
    the MMD/SUSHI families, whitespace-separated floats in brackets for the TSFragment families);
  * series are MinMax-scaled per COLUMN over the whole file before the split (dataset.py:81-82);
  * split = np.random.permutation(size) under np.random.seed(123), the first ceil(0.99 * size)
    rows are "train", the rest "test"; the global numpy RNG state is restored (dataset.py:43-69);
  * item = (text, x (L,), embedding (128,)) (dataset.py:98-102).
A parsed-once binary cache (`<name>.t2scache.npz` next to the CSV) replaces the per-row
ast.literal_eval on later runs.  `attach_latents` turns a dataset into a source of pre-encoded
LA-VAE latents (the encoder is frozen, train.py:31-33, yet re-run on every step,
train.py:106): items then carry their row index as a 4th element.
"""
from __future__ import annotations

import ast
import os

import numpy as np
from torch.utils.data import Dataset

_LITERAL_EMBEDDING_FAMILIES = {"Agriculture", "Climate", "Energy", "Health", "Security", "Traffic", "Economy",
                               "Environment", "SocialGood", "SUSHI"}


def split_indices(size: int, ratio: float = 0.99, seed: int = 123):
    """(train_idx, test_idx) exactly as T2SDataset.divide draws them (dataset.py:43-69)."""
    state = np.random.get_state()
    np.random.seed(seed)
    n_train = int(np.ceil(size * ratio))
    perm = np.random.permutation(size)
    np.random.set_state(state)
    return perm[:n_train], perm[n_train:]


def minmax_scale_columns(a: np.ndarray) -> np.ndarray:
    """sklearn MinMaxScaler().fit_transform semantics (feature_range (0,1), constant columns -> 0)."""
    # in sklearn's own operation order (scale_ = 1 / range, min_ = 0 - data_min * scale_, X * scale_ + min_), so the
    # result equals the reference's to the bit rather than to a few ulp of (a - lo) / span
    a = np.asarray(a, dtype=np.float64)
    lo, hi = np.nanmin(a, axis=0), np.nanmax(a, axis=0)
    span = hi - lo
    span[span < 10 * np.finfo(span.dtype).eps] = 1.0
    scale = 1.0 / span
    return a * scale + (0.0 - lo * scale)


class _LatentCacheMixin:
    """Row-index plumbing for the training latent cache (t2ms_amd/latent_cache.py)."""
    latents = None          # device tensor (N,64,w) once attached

    def attach_latents(self, latents):
        if latents.shape[0] != len(self):
            raise ValueError(f"latent cache has {latents.shape[0]} rows, dataset has {len(self)}")
        self.latents = latents
        return self

    def _item(self, i):
        base = (self.text[i], self.samples[i], self.embedding[i])
        return base if self.latents is None else base + (i,)


def _parse_embedding(cell, literal: bool) -> np.ndarray:
    if literal:
        return np.asarray(ast.literal_eval(cell), dtype=np.float64)
    return np.asarray([float(tok) for tok in cell.replace("[", " ").replace("]", " ").split()], dtype=np.float64)


def load_table(name: str, data_root: str):
    """-> (series float64 (N,L) MinMax-scaled, texts list[str], embeddings float64 (N,128))."""
    csv_path = os.path.join(data_root, name + ".csv")
    cache = os.path.join(data_root, name + ".t2scache.npz")
    if os.path.exists(cache) and os.path.getmtime(cache) >= os.path.getmtime(csv_path):
        z = np.load(cache, allow_pickle=True)
        return z["series"], list(z["texts"]), z["emb"]
    import pandas as pd
    df = pd.read_csv(csv_path)
    texts = df["Text"].tolist()
    raw = np.asarray([ast.literal_eval(c) if isinstance(c, str) else c for c in df["OT"]], dtype=np.float64)
    series = minmax_scale_columns(raw)
    literal = any(part in _LITERAL_EMBEDDING_FAMILIES for part in name.split("_"))
    emb = np.stack([_parse_embedding(c, literal) for c in df["TextEmbedding"]])
    try:
        np.savez(cache, series=series, texts=np.asarray(texts, dtype=object), emb=emb)
    except OSError:
        pass
    return series, texts, emb


class T2SDataset(_LatentCacheMixin, Dataset):
    def __init__(self, name="Agriculture", data_root="./Data/MMD", window=24, proportion=0.99, seed=123,
                 period="train", max_length=32):
        assert period in ("train", "test"), "period must be train or test."
        self.name, self.period, self.window, self.max_length = name, period, window, max_length
        series, texts, emb = load_table(name, data_root)
        if series.shape[0] != len(texts):
            raise ValueError("All inputs must have the same number of rows.")
        tr, te = split_indices(series.shape[0], proportion, seed)
        idx = tr if period == "train" else te
        self.samples = series[idx]
        self.text = [texts[i] for i in idx]
        self.embedding = emb[idx]
        self.len, self.var_num = series.shape[-1], 1
        self.sample_num = self.samples.shape[0]

    def __getitem__(self, i):
        return self._item(i)

    def __len__(self):
        return self.sample_num


class SyntheticT2SDataset(_LatentCacheMixin, Dataset):
    """Offline stand-in with the same item contract: U[0,1] series, unit-norm 128-d embeddings."""

    def __init__(self, n: int, length: int, seed: int = 2025):
        rs = np.random.RandomState(seed)
        self.samples = rs.uniform(0, 1, size=(n, length))
        e = rs.randn(n, 128)
        self.embedding = e / np.linalg.norm(e, axis=1, keepdims=True)
        self.text = [f"synthetic series {i}" for i in range(n)]
        self.len, self.var_num, self.sample_num = length, 1, n

    def __getitem__(self, i):
        return self._item(i)

    def __len__(self):
        return self.sample_num
