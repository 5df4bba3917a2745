"""Deterministic synthetic weights and inputs (no datasets / checkpoints offline).

Everything is drawn from ``numpy.random.RandomState`` (frozen stream) so the
same tensors can be regenerated on any box from a seed; golden fixtures store
only outputs.  State-dict keys and shapes are the reference's
(``model/denoiser/transformer.py:128-154``, ``model/pretrained/vqvae.py:36-95``,
``model/denoiser/mlp.py:49-94``; enumerated in SURVEY.md section 8b).

The default DiT initialiser zeroes the adaLN output linear
(``transformer.py:202-204``), which turns every block into the identity and
makes parity tests pass vacuously, so the synthetic weights draw it N(0, 0.02)
and use non-zero biases everywhere.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def _xavier(rs, out_f, in_f, gain=1.0):
    a = gain * math.sqrt(6.0 / (in_f + out_f))
    return rs.uniform(-a, a, size=(out_f, in_f)).astype(np.float32)


def _pos_embed(num_positions=480, d_model=128) -> torch.Tensor:
    # transformer.py:14-23 (fixed, non-trainable parameter kept in the state-dict)
    position = torch.arange(num_positions).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model)).unsqueeze(0)
    pe = torch.zeros(num_positions, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def make_dit_state_dict(seed: int = 2025, gain: float = 1.0, bias_std: float = 0.02,
                        adaln_std: float = 0.02) -> Dict[str, torch.Tensor]:
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f, g=gain):
        sd[name + ".weight"] = _xavier(rs, out_f, in_f, g)
        sd[name + ".bias"] = (rs.randn(out_f) * bias_std).astype(np.float32)

    sd["conv.weight"] = rs.uniform(-0.5, 0.5, size=(4, 1, 2, 2)).astype(np.float32)
    sd["conv.bias"] = rs.uniform(-0.5, 0.5, size=(4,)).astype(np.float32)
    lin("patch_emb", 128, 4)
    sd["ln.weight"] = (1.0 + 0.1 * rs.randn(128)).astype(np.float32)
    sd["ln.bias"] = (0.1 * rs.randn(128)).astype(np.float32)
    lin("linear_emb_to_patch", 4, 128)
    for i in range(4):
        p = f"layers.{i}."
        lin(p + "attn.qkv", 384, 128)
        lin(p + "attn.proj", 128, 128)
        lin(p + "mlp.fc1", 256, 128)
        lin(p + "mlp.fc2", 128, 256)
        sd[p + "adaLN_modulation.1.weight"] = (rs.randn(768, 128) * adaln_std).astype(np.float32)
        sd[p + "adaLN_modulation.1.bias"] = (rs.randn(768) * adaln_std).astype(np.float32)
    # dead-but-required keys (transformer.py:65-87,150)
    sd["unpatch.inv_embedding2d.weight"] = (rs.randn(128, 1, 6, 6) * 0.02).astype(np.float32)
    sd["unpatch.inv_embedding2d.bias"] = np.zeros(1, np.float32)
    lin("unpatch.fc1", 128, 60)
    lin("unpatch.fc2", 64, 128)
    out = {k: torch.from_numpy(v) for k, v in sd.items()}
    out["pos_embed"] = _pos_embed()
    return out


def make_vae_state_dict(seed: int = 2025, hidden: int = 128, n_res: int = 2, res_hidden: int = 256,
                        emb: int = 64) -> Dict[str, torch.Tensor]:
    """encoder.* and decoder.* of model/pretrained/vqvae.py at the default sizes
    (pretrained_lavae_unified.py:119-122)."""
    rs = np.random.RandomState(seed + 1)
    sd: Dict[str, np.ndarray] = {}

    def conv(name, out_c, in_c, k, bias=True, transposed=False):
        fan_in = in_c * k
        a = 1.0 / math.sqrt(fan_in)
        shape = (in_c, out_c, k) if transposed else (out_c, in_c, k)
        sd[name + ".weight"] = rs.uniform(-a, a, size=shape).astype(np.float32)
        if bias:
            sd[name + ".bias"] = rs.uniform(-a, a, size=(out_c,)).astype(np.float32)

    conv("encoder._conv_1", hidden // 2, 1, 4)
    conv("encoder._conv_2", hidden, hidden // 2, 4)
    conv("encoder._conv_3", hidden, hidden, 3)
    for i in range(n_res):
        conv(f"encoder._residual_stack._layers.{i}._block.1", res_hidden, hidden, 3, bias=False)
        conv(f"encoder._residual_stack._layers.{i}._block.3", hidden, res_hidden, 1, bias=False)
    conv("encoder._pre_vq_conv", emb, hidden, 1)
    conv("decoder._conv_1", hidden, emb, 3)
    for i in range(n_res):
        conv(f"decoder._residual_stack._layers.{i}._block.1", res_hidden, hidden, 3, bias=False)
        conv(f"decoder._residual_stack._layers.{i}._block.3", hidden, res_hidden, 1, bias=False)
    conv("decoder._conv_trans_1", hidden // 2, hidden, 4, transposed=True)
    conv("decoder._conv_trans_2", 1, hidden // 2, 4, transposed=True)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def make_mlp_state_dict(seed: int = 2025) -> Dict[str, torch.Tensor]:
    """model/denoiser/mlp.py:49-94 (8 layers; includes the constructed-but-unused
    norm1/norm3/pos_emb/self_attn/self_attn2 so load_state_dict(strict) passes)."""
    rs = np.random.RandomState(seed + 2)
    sd: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f):
        sd[name + ".weight"] = _xavier(rs, out_f, in_f)
        sd[name + ".bias"] = (rs.randn(out_f) * 0.02).astype(np.float32)

    for i in range(8):
        p = f"layers.{i}."
        for n, d in (("norm1", 64), ("norm2", 64), ("norm3", 6)):
            sd[p + n + ".weight"] = (1.0 + 0.1 * rs.randn(d)).astype(np.float32)
            sd[p + n + ".bias"] = (0.1 * rs.randn(d)).astype(np.float32)
        sd[p + "pos_emb.weight"] = rs.randn(12, 64).astype(np.float32)
        for n, d in (("self_attn", 64), ("self_attn2", 6)):
            sd[p + n + ".in_proj_weight"] = _xavier(rs, 3 * d, d)
            sd[p + n + ".in_proj_bias"] = np.zeros(3 * d, np.float32)
            sd[p + n + ".out_proj.weight"] = _xavier(rs, d, d)
            sd[p + n + ".out_proj.bias"] = np.zeros(d, np.float32)
        lin(p + "cross_attn.key", 64, 128)
        lin(p + "cross_attn.query", 64, 64)
        lin(p + "cross_attn.value", 64, 128)
        lin(p + "cross_attn.proj", 64, 64)
        lin(p + "mlp.0", 256, 64)
        lin(p + "mlp.2", 64, 256)
        lin(p + "mlp2.0", 256, 6)
        lin(p + "mlp2.2", 6, 256)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def make_latents(seed: int, batch: int, row0: int = 0) -> torch.Tensor:
    """x_T ~ N(0,1), (batch,64,30); row r is a function of (seed, row0+r) only."""
    out = np.empty((batch, 64, 30), np.float32)
    for r in range(batch):
        out[r] = np.random.RandomState((seed * 1000003 + row0 + r) % (2 ** 32)).randn(64, 30)
    return torch.from_numpy(out)


def make_text_embeddings(seed: int, batch: int, row0: int = 0) -> torch.Tensor:
    """Unit-norm N(0,1) rows, (batch,128): OpenAI text-embedding-3 outputs are L2-normalised
    and the reference's width is 128 (Get_Embedding_and_Convert_JSON_to_CSV.py:15-17)."""
    out = np.empty((batch, 128), np.float32)
    for r in range(batch):
        v = np.random.RandomState((seed * 998244353 + 7 + row0 + r) % (2 ** 32)).randn(128)
        out[r] = (v / np.linalg.norm(v)).astype(np.float32)
    return torch.from_numpy(out)


def make_series(seed: int, batch: int, length: int) -> torch.Tensor:
    """MinMax-scaled series stand-in: U[0,1], (batch, length) (dataset.py:81-82)."""
    rs = np.random.RandomState((seed + 11) % (2 ** 32))
    return torch.from_numpy(rs.uniform(0, 1, size=(batch, length)).astype(np.float32))


def make_ts2vec_state_dict(seed: int = 2025, input_dims: int = 1, output_dims: int = 100, hidden: int = 64,
                           depth: int = 10) -> Dict[str, torch.Tensor]:
    """Seeded weights under the state-dict keys of evaluate/ts2vec.py TSEncoder (:352-364: input_fc, then
    DilatedConvEncoder = depth blocks hidden->hidden + a final block hidden->output_dims with a 1x1 projector, :421-449).
    N(0, 1/fan_in) weights, small biases: activations stay O(1) through the 11 residual blocks."""
    rs = np.random.RandomState(seed)

    def w(*shape):
        fan_in = int(np.prod(shape[1:]))
        return torch.from_numpy((rs.randn(*shape) / np.sqrt(fan_in)).astype(np.float32))

    def b(n):
        return torch.from_numpy((0.05 * rs.randn(n)).astype(np.float32))

    sd = {"input_fc.weight": w(hidden, input_dims), "input_fc.bias": b(hidden)}
    chans = [hidden] * depth + [output_dims]
    for i, co in enumerate(chans):
        ci = chans[i - 1] if i > 0 else hidden
        p = f"feature_extractor.net.{i}."
        sd[p + "conv1.conv.weight"], sd[p + "conv1.conv.bias"] = w(co, ci, 3), b(co)
        sd[p + "conv2.conv.weight"], sd[p + "conv2.conv.bias"] = w(co, co, 3), b(co)
        if ci != co or i == len(chans) - 1:
            sd[p + "projector.weight"], sd[p + "projector.bias"] = w(co, ci, 1), b(co)
    return sd
