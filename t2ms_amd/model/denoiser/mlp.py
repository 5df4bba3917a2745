"""Mirror of the reference MLP denoiser (model/denoiser/mlp.py:49-94) -- BASELINE config 1 plumbing.

SURVEY.md 8(a) row a20 scopes this denoiser as plumbing only ("PyTorch is sufficient"): it is a
632 K-parameter model on a (B,64,6) latent that the reference itself can no longer run end to end
(its encoder emits 30 positions, vqvae.py:70).  So, unlike the DiT, this class evaluates with torch
ops on whatever device its tensors live on; it is NOT part of the accelerated path and no
throughput claim is made for it.  State-dict keys equal the reference's, including the modules it
constructs but never uses (norm1, norm3, pos_emb, self_attn, self_attn2).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

WIDTH, POSITIONS, TEXT_DIM, HEADS = 64, 6, 128, 4


def _time_features(t: torch.Tensor, dim: int) -> torch.Tensor:
    """mlp.py:5-18: [sin(100 t / f) | cos(100 t / f)], f = 10000**linspace(0,1,dim/2)."""
    f = torch.pow(10000, torch.linspace(0, 1, dim // 2)).to(t.device)
    arg = (t * 100.0).reshape(-1, 1) / f
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


class TimeEmbedding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0, "Dimension must be even"
        self.dim = dim

    def forward(self, t):
        return _time_features(t, self.dim)


class TextToSeriesCrossAttention(nn.Module):
    """mlp.py:21-47: series positions attend to the (repeated) text embedding."""

    def __init__(self, n_embd, condition_embd, n_head):
        super().__init__()
        assert n_embd % n_head == 0
        self.key = nn.Linear(condition_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(condition_embd, n_embd)
        self.proj = nn.Linear(n_embd, n_embd)
        self.n_head = n_head

    def forward(self, x, encoder_output, mask=None):
        B, T, C = x.shape
        hd = C // self.n_head
        split = lambda z: z.view(B, -1, self.n_head, hd).transpose(1, 2)      # noqa: E731
        q, k, v = split(self.query(x)), split(self.key(encoder_output)), split(self.value(encoder_output))
        att = F.softmax((q @ k.transpose(-2, -1)) / math.sqrt(hd), dim=-1)
        y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
        return self.proj(y), att.mean(dim=1)


class MLPlayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm1 = nn.LayerNorm(WIDTH)            # constructed, unused (mlp.py:53)
        self.norm2 = nn.LayerNorm(WIDTH)
        self.norm3 = nn.LayerNorm(POSITIONS)        # unused
        self.time_emb = TimeEmbedding(dim=WIDTH)
        self.pos_emb = nn.Embedding(POSITIONS * 2, embedding_dim=WIDTH, dtype=torch.float32)   # unused
        self.self_attn = nn.MultiheadAttention(WIDTH, 4)                                       # unused
        self.self_attn2 = nn.MultiheadAttention(POSITIONS, 2)                                  # unused
        self.cross_attn = TextToSeriesCrossAttention(WIDTH, TEXT_DIM, n_head=HEADS)
        self.mlp = nn.Sequential(nn.Linear(WIDTH, 256), nn.ReLU(), nn.Linear(256, WIDTH))
        self.mlp2 = nn.Sequential(nn.Linear(POSITIONS, 256), nn.ReLU(), nn.Linear(256, POSITIONS))

    def forward(self, input, t, text_input):
        """mlp.py:71-85: time-emb add -> cross-attn to text -> LN -> channel MLP -> position MLP."""
        h = (input + self.time_emb(t).unsqueeze(-1)).permute(0, 2, 1)          # (B,6,64)
        if text_input is not None:
            ctx = text_input.unsqueeze(1).expand(-1, POSITIONS, -1)
            h = h + self.cross_attn(h, ctx, ctx)[0]
        h = self.norm2(h)
        h = h + self.mlp(h)
        return self.mlp2(h.permute(0, 2, 1))


class MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([MLPlayer() for _ in range(8)])

    def forward(self, input, t, text_input):
        for layer in self.layers:
            input = layer(input, t, text_input)
        return input


for _cls in (MLP, MLPlayer, TextToSeriesCrossAttention, TimeEmbedding):
    _cls.__module__ = "model.denoiser.mlp"
