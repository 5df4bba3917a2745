"""TS2Vec for the C-FID metric (reference evaluate/ts2vec.py; evaluation.py:238-243): evaluation.py TRAINS this encoder on
the original series at evaluation time (`initialize_ts2vec`: 200 contrastive iterations from a random initialisation)
and then encodes original and generated series with it.

Split of labour here (the lowest-ranked remainder of the scope table):
  * `TS2Vec.encode(..., encoding_window='full_series')` -- what C-FID consumes -- runs the HIP encoder kernel
    (`t2s_ts2vec_encode`, csrc/t2s_eval.hip) on the averaged weights;
  * `TS2Vec.fit` is host-level plumbing like the MLP denoiser: a 0.6 M-parameter dilated-convolution stack trained for 200
    steps of batch 8.  Its forward / backward are torch autograd ops on the GPU; no throughput claim is made for it.

The random draws follow the reference's ORDER and SOURCES so that a run is reproducible against it under the same seeds:
module initialisation and the loader shuffle from torch's CPU generator, crops and binomial masks from numpy's global
generator (ts2vec.py:120-126, 349-350), and the dropout masks of the two forward passes from torch's CPU generator with
the shape `nn.Dropout` sees on a CPU run (B, C_out, T) -- on the GPU they are applied as a multiplication.
tests/golden/ts2vec_fit.npz holds the reference's loss curve and representations for the fixture.
"""
from __future__ import annotations

import copy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L


# ---------------------------------------------------------------------------- the encoder as a parameter container
class _SamePad(nn.Module):
    """ts2vec.py:401-418: Conv1d with 'same' length for any dilation (an even receptive field drops the last step)."""

    def __init__(self, cin, cout, k, dilation):
        super().__init__()
        rf = (k - 1) * dilation + 1
        self.conv = nn.Conv1d(cin, cout, k, padding=rf // 2, dilation=dilation)
        self.trim = 1 if rf % 2 == 0 else 0

    def forward(self, x):
        y = self.conv(x)
        return y[:, :, :-self.trim] if self.trim else y


class _Block(nn.Module):
    """ts2vec.py:420-433: GELU -> conv -> GELU -> conv, plus the (projected) input."""

    def __init__(self, cin, cout, k, dilation, final):
        super().__init__()
        self.conv1 = _SamePad(cin, cout, k, dilation)
        self.conv2 = _SamePad(cout, cout, k, dilation)
        self.projector = nn.Conv1d(cin, cout, 1) if (cin != cout or final) else None

    def forward(self, x):
        skip = x if self.projector is None else self.projector(x)
        return self.conv2(F.gelu(self.conv1(F.gelu(x)))) + skip


class _Dilated(nn.Module):
    def __init__(self, cin, channels, k):
        super().__init__()
        self.net = nn.Sequential(*[_Block(channels[i - 1] if i else cin, channels[i], k, 2 ** i, i == len(channels) - 1)
                                   for i in range(len(channels))])

    def forward(self, x):
        return self.net(x)


class TSEncoder(nn.Module):
    """ts2vec.py:352-399.  Same sub-module names (state-dict keys) and construction order (seeded initialisation) as the
    reference; `forward` is the TRAINING forward used by TS2Vec.fit (binomial mask, dropout mask handed in)."""

    def __init__(self, input_dims, output_dims, hidden_dims=64, depth=10):
        super().__init__()
        self.input_dims, self.output_dims, self.hidden_dims = input_dims, output_dims, hidden_dims
        self.input_fc = nn.Linear(input_dims, hidden_dims)
        self.feature_extractor = _Dilated(hidden_dims, [hidden_dims] * depth + [output_dims], 3)
        self.repr_dropout = nn.Dropout(p=0.1)

    def forward(self, x, mask, drop):
        """x (B,T,C_in) on the module's device, mask (B,T) bool (False = hidden time step), drop (B,C_out,T) the dropout
        multiplier (0 or 1/(1-p))."""
        valid = ~x.isnan().any(dim=-1)
        x = torch.where(valid.unsqueeze(-1), x, torch.zeros_like(x))
        h = self.input_fc(x)
        h = torch.where((mask & valid).unsqueeze(-1), h, torch.zeros_like(h))
        return (self.feature_extractor(h.transpose(1, 2)) * drop).transpose(1, 2)


# ---------------------------------------------------------------------------- loss (ts2vec.py:451-497)
def _pair_loss(sim, n):
    """-log-softmax of every row over all OTHER entries; mean over the positives (i, n + i) and (n + i, i)."""
    logits = torch.tril(sim, diagonal=-1)[..., :-1] + torch.triu(sim, diagonal=1)[..., 1:]
    logits = -F.log_softmax(logits, dim=-1)
    i = torch.arange(n, device=sim.device)
    return (logits[:, i, n + i - 1].mean() + logits[:, n + i, i].mean()) / 2


def instance_contrastive_loss(z1, z2):
    if z1.size(0) == 1:
        return z1.new_tensor(0.)
    z = torch.cat([z1, z2], dim=0).transpose(0, 1)                 # T x 2B x C
    return _pair_loss(torch.matmul(z, z.transpose(1, 2)), z1.size(0))


def temporal_contrastive_loss(z1, z2):
    if z1.size(1) == 1:
        return z1.new_tensor(0.)
    z = torch.cat([z1, z2], dim=1)                                  # B x 2T x C
    return _pair_loss(torch.matmul(z, z.transpose(1, 2)), z1.size(1))


def hierarchical_contrastive_loss(z1, z2, alpha=0.5, temporal_unit=0):
    loss, d = torch.zeros((), device=z1.device), 0
    while z1.size(1) > 1:
        if alpha != 0:
            loss = loss + alpha * instance_contrastive_loss(z1, z2)
        if d >= temporal_unit and 1 - alpha != 0:
            loss = loss + (1 - alpha) * temporal_contrastive_loss(z1, z2)
        d += 1
        z1 = F.max_pool1d(z1.transpose(1, 2), kernel_size=2).transpose(1, 2)
        z2 = F.max_pool1d(z2.transpose(1, 2), kernel_size=2).transpose(1, 2)
    if z1.size(1) == 1:
        if alpha != 0:
            loss = loss + alpha * instance_contrastive_loss(z1, z2)
        d += 1
    return loss / d


def _take_rows(x, starts, length):
    idx = torch.as_tensor(starts, device=x.device)[:, None] + torch.arange(length, device=x.device)[None, :]
    return x[torch.arange(x.size(0), device=x.device)[:, None], idx]


# ---------------------------------------------------------------------------- the model
class TS2Vec:
    """ts2vec.py:23-330 as far as evaluation.py uses it: constructor arguments, fit, encode('full_series'), save / load."""

    def __init__(self, input_dims, output_dims=320, hidden_dims=64, depth=10, device="cuda", lr=0.001, batch_size=16,
                 max_train_length=None, temporal_unit=0, after_iter_callback=None, after_epoch_callback=None):
        # fit() is torch autograd and runs wherever the module lives (a CPU run reproduces the reference's loss curve bit
        # for bit: tests/test_ts2vec_fit.py); encode() is the HIP kernel and needs a GPU (no CPU fallback)
        self.device = torch.device(device)
        self.lr, self.batch_size = lr, batch_size
        self.max_train_length, self.temporal_unit = max_train_length, temporal_unit
        self._net = TSEncoder(input_dims, output_dims, hidden_dims, depth).to(self.device)   # initialised on the CPU generator
        self.net = torch.optim.swa_utils.AveragedModel(self._net)
        self.net.update_parameters(self._net)
        self.after_iter_callback, self.after_epoch_callback = after_iter_callback, after_epoch_callback
        self.n_epochs = self.n_iters = 0
        self._hip = None

    # -- training (ts2vec.py:73-160)
    def fit(self, train_data, n_epochs=None, n_iters=None, verbose=False):
        from torch.utils.data import DataLoader, TensorDataset
        train_data = np.asarray(train_data)
        assert train_data.ndim == 3
        if n_iters is None and n_epochs is None:
            n_iters = 200 if train_data.size <= 100000 else 600
        if self.max_train_length is not None and train_data.shape[1] // self.max_train_length >= 2:
            raise L.T2SError("TS2Vec.fit: series longer than 2 x max_train_length (the reference's NaN-padded sectioning) "
                             "are outside what evaluation.py feeds it")
        if np.isnan(train_data).any():
            raise L.T2SError("TS2Vec.fit: missing values (NaN) are outside what evaluation.py feeds it")
        loader = DataLoader(TensorDataset(torch.from_numpy(train_data).to(torch.float)),
                            batch_size=min(self.batch_size, len(train_data)), shuffle=True, drop_last=True)
        opt = torch.optim.AdamW(self._net.parameters(), lr=self.lr)
        p_drop = self._net.repr_dropout.p
        self._hip = None
        loss_log = []
        while n_epochs is None or self.n_epochs < n_epochs:
            cum, n_in_epoch, stopped = 0.0, 0, False
            for (x,) in loader:
                if n_iters is not None and self.n_iters >= n_iters:
                    stopped = True
                    break
                if self.max_train_length is not None and x.size(1) > self.max_train_length:
                    off = np.random.randint(x.size(1) - self.max_train_length + 1)
                    x = x[:, off: off + self.max_train_length]
                x = x.to(self.device)
                T = x.size(1)
                crop_l = np.random.randint(low=2 ** (self.temporal_unit + 1), high=T + 1)
                left = np.random.randint(T - crop_l + 1)
                right = left + crop_l
                eleft = np.random.randint(left + 1)
                eright = np.random.randint(low=right, high=T + 1)
                offs = np.random.randint(low=-eleft, high=T - eright + 1, size=x.size(0))
                opt.zero_grad()
                outs = []
                for start, length in ((offs + eleft, right - eleft), (offs + left, eright - left)):
                    xs = _take_rows(x, start, length)
                    mask = torch.from_numpy(np.random.binomial(1, 0.5, size=(xs.size(0), xs.size(1)))).to(torch.bool)
                    keep = torch.empty(xs.size(0), self._net.output_dims, xs.size(1)).bernoulli_(1 - p_drop).div_(1 - p_drop)
                    outs.append(self._net(xs, mask.to(self.device), keep.to(self.device)))
                loss = hierarchical_contrastive_loss(outs[0][:, -crop_l:], outs[1][:, :crop_l], temporal_unit=self.temporal_unit)
                loss.backward()
                opt.step()
                self.net.update_parameters(self._net)
                val = loss.item()
                cum += val
                n_in_epoch += 1
                self.n_iters += 1
                if self.after_iter_callback is not None:
                    self.after_iter_callback(self, val)
            if stopped:
                break
            loss_log.append(cum / n_in_epoch)
            if verbose:
                print(f"Epoch #{self.n_epochs}: loss={loss_log[-1]}")
            self.n_epochs += 1
            if self.after_epoch_callback is not None:
                self.after_epoch_callback(self, loss_log[-1])
        return loss_log

    # -- inference on the HIP kernel (ts2vec.py:219-330, the 'full_series' / per-step windows)
    def _encoder(self):
        if self._hip is None:
            from .metrics import TS2VecEncoder
            self._hip = TS2VecEncoder(copy.deepcopy(self.net.module.state_dict()), self.device)
        return self._hip

    def encode(self, data, encoding_window=None, batch_size=None):
        data = np.asarray(data)
        assert data.ndim == 3
        bs = batch_size or self.batch_size
        enc = self._encoder()
        outs = [enc.encode(torch.from_numpy(data[i:i + bs]).float(), encoding_window=encoding_window).cpu()
                for i in range(0, len(data), bs)]
        return torch.cat(outs, dim=0).numpy()

    def save(self, fn):
        torch.save(self.net.state_dict(), fn)

    def load(self, fn):
        self.net.load_state_dict(torch.load(fn, map_location=self.device))
        self._hip = None


def initialize_ts2vec(X_train, device="cuda"):
    """ts2vec.py:12-21: the configuration evaluation.py:238 trains for C-FID."""
    model = TS2Vec(input_dims=X_train.shape[-1], device=device, batch_size=8, lr=0.001, output_dims=100, max_train_length=3000)
    model.fit(X_train, verbose=False)
    return model
