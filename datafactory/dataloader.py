"""loader_provider(args, period) -> (dataset, DataLoader)   (reference datafactory/dataloader.py:79-133).

Kept: dataset-name -> CSV-name mapping, the three data roots, mix-train = three lengths
(24/48/96) concatenated behind one loader whose collate groups a batch by length, and
`shuffle=True, drop_last=True` for BOTH periods (evaluation.py relies on N = floor(test/B)*B).
New: `args.synthetic` (int rows) serves synthetic data when the CSVs are not on disk.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset

from .dataset import SyntheticT2SDataset, T2SDataset

_FRAGMENT = ("ETTh1", "ETTm1", "traffic", "airquality", "exchangerate", "weather", "electricity", "nationalillness")
_MMD = ("Agriculture", "Climate", "Health_US", "Traffic", "Economy", "SocialGood")


def csv_name(dataset_name: str) -> str:
    """'ETTh1_24' -> 'embedding_cleaned_ETTh1_24', 'MMD-Climate_48' -> 'embedding_cleaned_Climate_48', ..."""
    base = dataset_name.split("-", 1)[1] if dataset_name.startswith("MMD-") else dataset_name
    return "embedding_cleaned_" + base


data_dict = {n: csv_name(n) for n in
             [f"{b}{s}" for b in _FRAGMENT[:6] + ("electricity",) for s in ("", "_24", "_48", "_96")] +
             [f"MMD-{b}{s}" for b in _MMD for s in ("", "_24", "_48", "_96")] + ["SUSHI"]}


class AlternatingDataset(Dataset):
    """Three datasets back to back; item = (sample, which_dataset) (dataloader.py:6-22)."""

    def __init__(self, d1, d2, d3):
        self.datasets = [d1, d2, d3]
        self._cat = ConcatDataset(self.datasets)
        self._starts = np.cumsum([0] + [len(d) for d in self.datasets])

    def __len__(self):
        return len(self._cat)

    def __getitem__(self, i):
        which = int(np.searchsorted(self._starts, i, side="right") - 1)
        return self._cat[i], which


def custom_collate_fn(batch):
    """Group a mixed batch by source dataset -> list of (texts, xs, embeddings) (dataloader.py:115-133);
    with a latent cache attached each group carries its dataset row indices as a 4th element."""
    out = []
    for which in (0, 1, 2):
        rows = [item for item, w in batch if w == which]
        if not rows:
            continue
        cols = list(zip(*rows))
        group = (list(cols[0]), torch.stack([torch.as_tensor(x) for x in cols[1]]),
                 torch.stack([torch.as_tensor(e) for e in cols[2]]))
        if len(cols) == 4:
            group += (torch.as_tensor(cols[3], dtype=torch.long),)
        out.append(group)
    return out


def _root_for(dataset_name: str, mix: bool) -> str:
    family = dataset_name.split("_")[0]
    if family in _FRAGMENT:
        return "./Data/TSFragment-600K/" if mix else "./Data/our/"
    if dataset_name == "SUSHI":
        return "./Data/SUSHI/"
    if dataset_name.startswith("MMD-"):
        return "./Data/MMD/"
    raise ValueError(f"unknown dataset {dataset_name!r}")


def loader_provider(args, period):
    synthetic = int(getattr(args, "synthetic", 0) or 0)
    name = args.dataset_name
    if getattr(args, "mix_train", False):
        if synthetic:
            parts = [SyntheticT2SDataset(synthetic, L, seed=2025 + L) for L in (24, 48, 96)]
        elif name == "SUSHI":
            ds = T2SDataset(name=csv_name(name), data_root=_root_for(name, True), period=period)
            return ds, DataLoader(ds, batch_size=args.batch_size, shuffle=True, drop_last=True,
                                  collate_fn=custom_collate_fn)
        else:
            root = _root_for(name, True)
            parts = [T2SDataset(name=f"{csv_name(name)}_{L}", data_root=root, period=period) for L in (24, 48, 96)]
        ds = AlternatingDataset(*parts)
        return ds, DataLoader(ds, batch_size=args.batch_size, shuffle=True, drop_last=True,
                              collate_fn=custom_collate_fn)
    if synthetic:
        length = int(name.rsplit("_", 1)[1]) if "_" in name and name.rsplit("_", 1)[1].isdigit() else 96
        ds = SyntheticT2SDataset(synthetic, length)
    else:
        ds = T2SDataset(name=csv_name(name), data_root=_root_for(name, False), period=period)
    return ds, DataLoader(ds, batch_size=args.batch_size, shuffle=True, drop_last=True)


# ------------------------------------------------------------------ the loader's ORDER without the loader's per-row work
def epoch_index_batches(loader) -> torch.Tensor:
    """The index batches ONE `for data in loader` pass would visit, as an (n_batches, batch_size) int64 tensor, consuming
    the global CPU generator exactly as that pass does -- so a driver may mix this with real passes over the same loader.

    A single-process `DataLoader(shuffle=True, drop_last=True)` (dataloader.py:99,111 of the reference) draws, per pass:
    one int64 from the global generator when the iterator is built (its `_base_seed`), one more when the first batch is
    asked for (`RandomSampler.__iter__`: the seed of a private generator), then `torch.randperm(n)` from that private
    generator; batch k is rows perm[k * B : (k + 1) * B].  No dataset row is touched here: the drivers gather rows from
    resident tensors instead of running `__getitem__` + collate per row (tests/test_host_logic.py pins this against a
    real pass)."""
    n, bs = len(loader.dataset), int(loader.batch_size)
    if loader.generator is not None or getattr(loader.sampler, "generator", None) is not None:
        # with an explicit generator RandomSampler also draws a trailing randperm from it: not mirrored here -- walk the
        # loader instead (walk_index_batches) rather than emulate a shape the T2S drivers never build
        raise ValueError("epoch_index_batches: loaders with an explicit generator are not supported; use walk_index_batches")
    torch.empty((), dtype=torch.int64).random_()                                    # _BaseDataLoaderIter._base_seed
    g = torch.Generator()
    g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))         # RandomSampler.__iter__
    perm = torch.randperm(n, generator=g)
    nb = n // bs if loader.drop_last else -(-n // bs)
    if nb * bs > n:          # drop_last=False with a ragged tail: not a shape the T2S drivers build
        raise ValueError("epoch_index_batches: ragged last batch (drop_last=False) is not supported")
    return perm[: nb * bs].view(nb, bs)


class _RowIndex(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def walk_index_batches(loader) -> torch.Tensor:
    """The same (n_batches, batch_size) index batches obtained by WALKING a DataLoader of the same shape over the row numbers
    -- public torch API only, no knowledge of how DataLoader / RandomSampler draw.  The draws of a pass do not depend on what
    the dataset returns, so this consumes the CPU generator exactly as a pass over `loader` does; it costs one Python call per
    row (0.2 s per 600 K rows), which is why the drivers default to epoch_index_batches and keep this as the cross-check
    (`--loader_batches`; tests/test_host_logic.py holds the two and a real pass equal)."""
    if not loader.drop_last and len(loader.dataset) % int(loader.batch_size):
        raise ValueError("walk_index_batches: ragged last batch (drop_last=False) is not supported")
    shuffle = isinstance(loader.sampler, torch.utils.data.RandomSampler)
    twin = DataLoader(_RowIndex(len(loader.dataset)), batch_size=int(loader.batch_size), shuffle=shuffle,
                      drop_last=loader.drop_last, generator=loader.generator)
    rows = [b for b in twin]
    if not rows:
        return torch.empty(0, int(loader.batch_size), dtype=torch.int64)
    return torch.stack(rows).to(torch.int64)


def resident_tables(dataset):
    """-> list of (series float64 (N,L) ndarray, embeddings float64 (N,128) ndarray, first index in the concatenation) per
    leaf dataset, in AlternatingDataset order (24, 48, 96) or a single entry for a plain dataset."""
    leaves = dataset.datasets if hasattr(dataset, "datasets") else [dataset]
    out, start = [], 0
    for d in leaves:
        out.append((d.samples, d.embedding, start))
        start += len(d)
    return out


def group_by_dataset(idx: torch.Tensor, starts) -> list:
    """custom_collate_fn's grouping (dataloader.py:115-133) on an index batch: [(which, rows within that dataset)] in
    dataset order, each group's rows in batch order; empty groups are skipped."""
    bounds = torch.as_tensor(list(starts[1:]), dtype=torch.int64)
    which = torch.bucketize(idx, bounds, right=True)
    out = []
    for w in range(len(starts)):
        sel = idx[which == w]
        if sel.numel():
            out.append((w, sel - int(starts[w])))
    return out


def plan_epochs(loader, starts, mix: bool, first_epoch: int, last_epoch: int, min_steps: int = 64, coin_p: float = 0.3):
    """Index batches of SEVERAL epochs at once for a resident training loop (train.py of this repository).

    For epochs first_epoch, first_epoch + 1, ... (< last_epoch) until the plan holds >= min_steps optimisation steps, the
    global CPU generator is consumed exactly as a loader pass + the steps' classifier-free-guidance coins consume it (two
    draws when the pass starts -- epoch_index_batches -- then ONE `torch.rand(1) < coin_p` per length group in visiting
    order, train.py:120-122 of the reference), the rows of each batch are put in custom_collate_fn's group order (stable:
    dataset 0's rows first, each group in batch order) and made local to their dataset.

    -> (plans, flat): plans = [(epoch, rows (nb, B) int64, groups [[(dataset, count)]], coins [[bool]])]; flat = every
    plan's rows concatenated (ONE host -> device copy serves all of its steps: batch b of plan p starts at
    sum(rows.numel() of earlier plans) + b * B, its groups follow each other)."""
    plans, n_steps, e = [], 0, first_epoch
    starts_t = torch.as_tensor(list(starts), dtype=torch.int64)
    while e < last_epoch and (not plans or n_steps < min_steps):
        batches = epoch_index_batches(loader)                                       # (nb, B) rows of the concatenation
        if mix and len(starts) > 1:
            which = torch.bucketize(batches, starts_t[1:], right=True)
            order = torch.argsort(which, dim=1, stable=True)
            which = which.gather(1, order)
            rows = batches.gather(1, order) - starts_t[which]
            counts = torch.stack([(which == w).sum(1) for w in range(len(starts))], dim=1)        # (nb, groups)
        else:
            rows, counts = batches, torch.full((batches.shape[0], 1), batches.shape[1], dtype=torch.int64)
        groups = [[(w, int(c)) for w, c in enumerate(row) if c] for row in counts.tolist()]
        coins = [[bool(torch.rand(1) < coin_p) for _ in g] for g in groups]
        plans.append((e, rows, groups, coins))
        n_steps += sum(len(g) for g in groups)
        e += 1
    flat = torch.cat([p[1].reshape(-1) for p in plans]) if plans else torch.empty(0, dtype=torch.int64)
    return plans, flat
