"""Pre-encoded LA-VAE latents for DiT training.

The reference freezes the encoder (train.py:31-33) and still runs it on every batch of every epoch
(train.py:66,106).  Its output depends on the series alone, so each dataset is encoded ONCE on the
GPU (t2s_vae_encode in chunks) and a training step gathers its rows from the resident tensor:
600 K rows x 64 x 30 fp32 = 4.6 GB of the 288 GB HBM.  Results are bit-identical to re-encoding.
"""
from __future__ import annotations

import torch
from torch.utils.data import ConcatDataset


def leaf_datasets(dataset):
    """The datasets that own `samples` (unwraps AlternatingDataset / ConcatDataset)."""
    if hasattr(dataset, "datasets"):
        out = []
        for d in dataset.datasets:
            out += leaf_datasets(d)
        return out
    if isinstance(dataset, ConcatDataset):
        return [x for d in dataset.datasets for x in leaf_datasets(d)]
    return [dataset]


@torch.no_grad()
def encode_all(encoder, samples, device, chunk: int = 4096) -> torch.Tensor:
    """samples (N,L) array-like -> latents (N,64,30) fp32 on `device` (encoder = vqvae.Encoder mirror)."""
    x = torch.as_tensor(samples)
    out = []
    for lo in range(0, x.shape[0], chunk):
        z, _ = encoder(x[lo:lo + chunk].float().to(device).contiguous())
        out.append(z)
    return torch.cat(out, dim=0)


def attach(dataset, encoder, device) -> dict:
    """Encode every leaf dataset and attach the latents; returns {series length: latents}."""
    by_len = {}
    for d in leaf_datasets(dataset):
        d.attach_latents(encode_all(encoder, d.samples, device))
        by_len[int(d.len)] = d.latents
    return by_len
