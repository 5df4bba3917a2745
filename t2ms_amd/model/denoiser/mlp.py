"""Mirror of the reference MLP denoiser (model/denoiser/mlp.py:49-94) -- BASELINE configs[0].

A 632 K-parameter model on a (B,64,6) latent that the reference itself can no longer run end to end (its encoder emits 30
positions, vqvae.py:70; SURVEY.md 8(d) prescribes the runnable form).  SURVEY 8(a) row a20 scopes it as plumbing.
  * Inference on a GPU (no autograd): `MLP.forward` is ONE launch of the HIP kernel `t2s_mlp_forward` (csrc/t2s_mlp.hip:
    all eight layers, one workgroup per series) on weights packed by `t2s_mlp_pack`; there is no other GPU inference path
    (a missing library raises).
  * Under autograd on a GPU (train.py --denoiser MLP) the same forward runs and the backward is `t2s_mlp_backward` (one
    autograd node, `_MlpFn`; T2S_MLP_TORCH_AUTOGRAD=1 keeps torch-op autograd for A/B).
  * On CPU tensors the layers below evaluate with torch ops, as the reference (the tests' reference path).
State-dict keys equal the reference's, including the modules it constructs but never uses (norm1, norm3, pos_emb,
self_attn, self_attn2) and the cross attention's query / key, which cannot influence the result (see t2s.h).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib as L

WIDTH, POSITIONS, TEXT_DIM, HEADS = 64, 6, 128, 4


def _time_features(t: torch.Tensor, dim: int) -> torch.Tensor:
    """mlp.py:5-18: [sin(100 t / f) | cos(100 t / f)], f = 10000**linspace(0,1,dim/2)."""
    f = torch.pow(10000, torch.linspace(0, 1, dim // 2)).to(t.device)
    arg = (t * 100.0).reshape(-1, 1) / f
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


_FREQS = {}


def _freqs_on(device):
    """10000**linspace(0,1,32) evaluated by the reference's own fp32 torch ops on the host (mlp.py:12), once per device."""
    f = _FREQS.get(device)
    if f is None:
        f = _FREQS[device] = torch.pow(10000, torch.linspace(0, 1, WIDTH // 2)).to(device)
    return f


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
        needs_grad = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))
        if input.is_cuda and not needs_grad:
            return self._forward_hip(input, t, text_input)
        if (input.is_cuda and len(self.layers) == L.MLP_LAYERS and tuple(input.shape[1:]) == (WIDTH, POSITIONS)
                and os.environ.get("T2S_MLP_TORCH_AUTOGRAD", "0") in ("", "0")):
            # training on a GPU: forward AND backward in the HIP kernels (t2s_mlp_forward / t2s_mlp_backward)
            return _MlpFn.apply(self, input, t, text_input, *self._all_layer_params())
        for layer in self.layers:
            input = layer(input, t, text_input)
        return input

    def _all_layer_params(self):
        """The 14 tensors per layer the kernels read (t2s_mlp_layer_weights order), then cross_attn.query / key weight and bias
        per layer: they cannot influence the forward, autograd hands them exact zeros."""
        qk = []
        for layer in self._modules["layers"]._modules.values():
            ca = layer._modules["cross_attn"]._modules
            qk += [ca["query"]._parameters["weight"], ca["query"]._parameters["bias"], ca["key"]._parameters["weight"],
                   ca["key"]._parameters["bias"]]
        return self._hip_tensors() + qk

    # -- the HIP path -------------------------------------------------------------------------------------------------
    def _hip_tensors(self):
        """The 14 tensors per layer t2s_mlp_pack reads, in t2s_mlp_layer_weights order.  Walks the modules' own dicts
        (dict(named_parameters()) costs 0.3 ms per call, more than the kernel); nothing is cached, so a re-assigned
        Parameter is still seen."""
        ts = []
        for layer in self._modules["layers"]._modules.values():
            m = layer._modules
            ca, n2, mlp, mlp2 = m["cross_attn"]._modules, m["norm2"]._parameters, m["mlp"]._modules, m["mlp2"]._modules
            for holder in (ca["value"]._parameters, ca["proj"]._parameters, n2, mlp["0"]._parameters, mlp["2"]._parameters,
                           mlp2["0"]._parameters, mlp2["2"]._parameters):
                ts.append(holder["weight"])
                ts.append(holder["bias"])
        return ts

    def _packed(self, device):
        """The transposed weights `t2s_mlp_forward` reads, re-packed whenever a parameter's storage or in-place version
        changed (optimizer steps, load_state_dict, .to()); never cached for parameters torch keeps no version for."""
        from .transformer import _stamp_of
        ts = self._hip_tensors()
        stamp = _stamp_of(ts)
        cached = self.__dict__.get("_t2s_packed")
        if cached is not None and stamp is not None and cached[0] == device and cached[1] == stamp:
            return cached[2]
        if len(self.layers) != L.MLP_LAYERS:
            raise L.T2SError(f"t2s_mlp_forward is built for {L.MLP_LAYERS} layers, this model has {len(self.layers)}")
        keep = []
        w = L.MlpWeights()
        it = iter(ts)
        for i in range(L.MLP_LAYERS):
            for field, key in L.MLP_LAYER_FIELDS:
                t = next(it)
                if t.device != device:
                    raise L.T2SError(f"MLP parameters live on {t.device} but the input is on {device}; call model.to(device) first")
                t = L.as_f32(t.detach())
                keep.append(t)
                setattr(w.layer[i], field, L.dev_ptr(t, f"layers.{i}.{key}"))
        packed = torch.empty(L.MLP_PACKED_FLOATS, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            L.check(L.lib().t2s_mlp_pack(w, packed.data_ptr(), L.stream_ptr(device)), "t2s_mlp_pack")
        self.__dict__["_t2s_packed"] = (device, stamp, packed)
        return packed

    def _forward_hip(self, input, t, text_input):
        if input.dim() != 3 or input.shape[1] != WIDTH or input.shape[2] != POSITIONS:
            raise L.T2SError(f"MLP.forward: input must be (B,{WIDTH},{POSITIONS}), got {tuple(input.shape)}")
        B, device = input.shape[0], input.device
        if text_input is not None and tuple(text_input.shape) != (B, TEXT_DIM):
            raise L.T2SError(f"MLP.forward: text_input must be ({B},{TEXT_DIM}), got {tuple(text_input.shape)}")
        if t.numel() != B:
            raise L.T2SError(f"MLP.forward: t must hold {B} values, got {tuple(t.shape)}")
        packed = self._packed(device)
        x = L.as_f32(input)
        tf = L.as_f32(t.to(device).reshape(-1))                                   # `t * 100.0` promotes an int64 t the same way
        text = None if text_input is None else L.as_f32(text_input.to(device))
        out = torch.empty_like(x)
        with torch.cuda.device(device):
            L.check(L.lib().t2s_mlp_forward(packed.data_ptr(), L.dev_ptr(x, "input"), L.dev_ptr(tf, "t"),
                                            L.dev_ptr(_freqs_on(device)), L.dev_ptr(text, "text_input"), out.data_ptr(), B,
                                            L.stream_ptr(device)), "t2s_mlp_forward")
        return out


class _MlpFn(torch.autograd.Function):
    """MLP.forward under autograd with both directions in the HIP kernels.  Only the input, t and the text are saved: the
    backward kernel recomputes the forward (the model is 6.5 MFLOP per series)."""

    @staticmethod
    def forward(ctx, model, input, t, text_input, *params):
        with torch.no_grad():
            out = model._forward_hip(input, t, text_input)
        ctx.model = model
        ctx.has_text = text_input is not None
        ctx.save_for_backward(L.as_f32(input), L.as_f32(t.to(input.device).reshape(-1)),
                              L.as_f32(text_input.to(input.device)) if text_input is not None else input.new_zeros(1))
        ctx.n_params = len(params)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, tf, text = ctx.saved_tensors
        model, dev, B = ctx.model, x.device, x.shape[0]
        packed = model._packed(dev)                                   # the forward's weights: no step happens in between
        ts = model._hip_tensors()
        keep, w, g = [], L.MlpWeights(), L.MlpGrads()
        grads = [torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in ts]
        it, ig = iter(ts), iter(grads)
        for i in range(L.MLP_LAYERS):
            for field, _ in L.MLP_LAYER_FIELDS:
                src = L.as_f32(next(it).detach())
                keep.append(src)
                setattr(w.layer[i], field, L.dev_ptr(src))
                setattr(g.layer[i], field, next(ig).data_ptr())
        need_dx = ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_dx else None
        scratch = torch.empty(B * L.MLP_GRAD_PART_FLOATS, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().t2s_mlp_backward(C.byref(w), packed.data_ptr(), L.dev_ptr(x), L.dev_ptr(tf), L.dev_ptr(_freqs_on(dev)),
                                             L.dev_ptr(text) if ctx.has_text else None, L.dev_ptr(L.as_f32(dout)),
                                             None if dx is None else dx.data_ptr(), C.byref(g), scratch.data_ptr(), scratch.numel(), B,
                                             L.stream_ptr(dev)), "t2s_mlp_backward")
        # without a text the cross attention is skipped (mlp.py:75): its parameters get NO gradient, as under torch autograd
        n_f = len(L.MLP_LAYER_FIELDS)
        out = [gr if p.requires_grad and (ctx.has_text or i % n_f >= 4) else None for i, (gr, p) in enumerate(zip(grads, ts))]
        qk = [torch.zeros_like(p) if p.requires_grad and ctx.has_text else None for p in model._all_layer_params()[len(ts):]]
        return (None, dx, None, None, *out, *qk)


for _cls in (MLP, MLPlayer, TextToSeriesCrossAttention, TimeEmbedding):
    _cls.__module__ = "model.denoiser.mlp"
