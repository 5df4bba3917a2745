"""CPU oracle for the T2S diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain fp32 PyTorch-CPU restatement of the reference algorithm
(Bill9125/T2MS) for the path named by BASELINE.json:north_star.  It is the
*checker* for the HIP product path in ``t2ms_amd/``; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing in ``t2ms_amd/`` imports it, and it never runs on behalf
of the product.

Parity status
-------------
* DDPM / RectifiedFlow / LA-VAE / TimeEmbedding / patchify / adaLN wiring /
  unpatchify / MLP denoiser: PINNED.  ``tests/golden/gen_golden.py`` imported
  the real reference modules in the build container and the outputs are
  committed under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
  checks this file against them.
* ``timm.models.vision_transformer.Attention`` / ``Mlp`` (timm==1.0.11,
  reference ``requirements.txt:9``): the source of that dependency is NOT in
  /root/reference and timm is not installed, so their arithmetic is restated
  here from the published timm 1.0.11 algorithm and anchored only on the
  reference call sites (``model/denoiser/transformer.py:104-105,116-117``).
  The reference has no tests or golden vectors for this boundary:
  **parity unpinned at the timm boundary**.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  All functions are functional over a state-dict whose keys
are the reference's own.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

D_MODEL = 128
N_HEADS = 4
HEAD_DIM = 32
N_TOK = 480
N_BLOCKS = 4
LAT_C = 64
LAT_W = 30


# --------------------------------------------------------------------------
# DDPM  (model/backbone/DDPM.py)
# --------------------------------------------------------------------------
def ddpm_tables(total_steps: int) -> Dict[str, Tensor]:
    """DDPM.__init__, model/backbone/DDPM.py:11-18."""
    beta = torch.linspace(0.0001, 0.02, total_steps)
    alpha = 1 - beta
    alpha_bar = torch.cumprod(alpha, dim=0)
    return dict(beta=beta, alpha=alpha, alpha_bar=alpha_bar, sigma2=beta)


def _gather(consts: Tensor, t: Tensor) -> Tensor:
    """gather, model/backbone/DDPM.py:7-9."""
    return consts.gather(-1, t).reshape(-1, 1, 1)


def ddpm_q_sample(tab, x0: Tensor, t: Tensor, eps: Tensor) -> Tensor:
    """q_xt_x0 + q_sample with injected eps, model/backbone/DDPM.py:19-27."""
    mean = _gather(tab["alpha_bar"], t) ** 0.5 * x0
    var = 1 - _gather(tab["alpha_bar"], t)
    return mean + (var ** 0.5) * eps


def ddpm_p_sample(tab, xt: Tensor, eps_hat: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """p_sample with the Gaussian draw injected, model/backbone/DDPM.py:28-36.

    The reference draws ``torch.randn`` at every step INCLUDING t=0.
    """
    alpha_bar = _gather(tab["alpha_bar"], t)
    alpha = _gather(tab["alpha"], t)
    eps_coef = (1 - alpha) / (1 - alpha_bar) ** .5
    mean = 1 / (alpha ** 0.5) * (xt - eps_coef * eps_hat)
    var = _gather(tab["sigma2"], t)
    return mean + (var ** .5) * noise


def mse_loss(a: Tensor, b: Tensor) -> Tensor:
    """DDPM.loss / RectifiedFlow.loss, DDPM.py:37-38, rectified_flow.py:13-16."""
    return F.mse_loss(a, b)


# --------------------------------------------------------------------------
# Rectified flow  (model/backbone/rectified_flow.py)
# --------------------------------------------------------------------------
def rf_euler(x_t: Tensor, v: Tensor, dt: float) -> Tensor:
    """RectifiedFlow.euler, model/backbone/rectified_flow.py:5-7."""
    return x_t + v * dt


def rf_create_flow(x_1: Tensor, t: Tensor, x_0: Tensor) -> Tensor:
    """RectifiedFlow.create_flow with x_0 injected, rectified_flow.py:8-12."""
    t = t[:, None, None]
    return t * x_1 + (1 - t) * x_0


# --------------------------------------------------------------------------
# DiT pieces  (model/denoiser/transformer.py)
# --------------------------------------------------------------------------
def time_freqs(dim: int = D_MODEL) -> Tensor:
    """freqs of TimeEmbedding.forward, transformer.py:34."""
    return torch.pow(10000, torch.linspace(0, 1, dim // 2))


def time_embedding(t: Tensor, dim: int = D_MODEL) -> Tensor:
    """TimeEmbedding.forward, transformer.py:30-40.  t is (B,) int64 or float."""
    t = t * 100.0
    t = t.unsqueeze(-1)
    freqs = time_freqs(dim)
    arg = t[:, None] / freqs
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
    return emb.squeeze(1)


def sinusoidal_pos_embed(num_positions: int = N_TOK, d_model: int = D_MODEL) -> Tensor:
    """get_sinusoidal_positional_embeddings, transformer.py:14-23."""
    position = torch.arange(num_positions).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model)).unsqueeze(0)
    pe = torch.zeros(num_positions, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """modulate, transformer.py:7-8."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


_ATTENTION_IMPL = "explicit"


def set_attention_impl(impl: str) -> None:
    """Which of timm 1.0.11 Attention.forward's two branches timm_attention follows: "explicit" (the non-fused
    branch, the default here: every intermediate is inspectable) or "sdpa" (F.scaled_dot_product_attention, the branch
    timm takes when `fused_attn` is available, i.e. what the reference executes; bench.py times this one as the CPU
    baseline).  The two agree to fp32 rounding (tests/test_oracle_golden.py)."""
    global _ATTENTION_IMPL
    if impl not in ("explicit", "sdpa"):
        raise ValueError(impl)
    _ATTENTION_IMPL = impl


def timm_attention(x: Tensor, w_qkv: Tensor, b_qkv: Tensor, w_proj: Tensor, b_proj: Tensor,
                   num_heads: int = N_HEADS) -> Tensor:
    """timm 1.0.11 vision_transformer.Attention.forward (restated; unpinned).

    Call site transformer.py:104,116: Attention(128, num_heads=4, qkv_bias=True);
    q_norm/k_norm are Identity, dropouts are 0.  Default: the explicit (non-fused)
    form softmax((q*scale) @ k^T) @ v; see set_attention_impl for the fused branch.
    """
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, w_qkv, b_qkv).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if _ATTENTION_IMPL == "sdpa":
        o = F.scaled_dot_product_attention(q, k, v)
    else:
        q = q * (hd ** -0.5)
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        o = attn @ v
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, w_proj, b_proj)


def timm_mlp(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """timm 1.0.11 layers.Mlp.forward with act=GELU(tanh) (restated; unpinned).

    Call site transformer.py:100,105,117.
    """
    return F.linear(F.gelu(F.linear(x, w1, b1), approximate="tanh"), w2, b2)


def dit_patchify(sd: SD, inp: Tensor) -> Tensor:
    """transformer.py:166-172: permute, Conv2d 1->4 2x2 s2, Linear 4->128, +pos."""
    x = inp.permute(0, 2, 1).unsqueeze(1)
    x = F.conv2d(x, sd["conv.weight"], sd["conv.bias"], stride=2)
    x = x.permute(0, 2, 3, 1)
    x = x.reshape(x.size(0), N_TOK, x.size(3))
    x = F.linear(x, sd["patch_emb.weight"], sd["patch_emb.bias"])
    return x + sd["pos_embed"]


def dit_block(sd: SD, i: int, x: Tensor, c: Tensor, taps: Optional[dict] = None) -> Tensor:
    """Transformerlayer.forward, transformer.py:114-117."""
    p = f"layers.{i}."
    mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
    h = F.layer_norm(x, (D_MODEL,), eps=1e-6)
    x = x + g1.unsqueeze(1) * timm_attention(
        modulate(h, sh1, sc1), sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    if taps is not None:
        taps[f"post_attn_{i}"] = x
    h = F.layer_norm(x, (D_MODEL,), eps=1e-6)
    x = x + g2.unsqueeze(1) * timm_mlp(
        modulate(h, sh2, sc2), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"],
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if taps is not None:
        taps[f"post_mlp_{i}"] = x
    return x


def dit_unpatchify(sd: SD, x: Tensor) -> Tensor:
    """transformer.py:182-191: affine LN (eps 1e-5), Linear 128->4, unpatchify."""
    x = F.layer_norm(x, (D_MODEL,), sd["ln.weight"], sd["ln.bias"], eps=1e-5)
    x = F.linear(x, sd["linear_emb_to_patch.weight"], sd["linear_emb_to_patch.bias"])
    B = x.size(0)
    x = x.view(B, 15, 32, 1, 2, 2).permute(0, 3, 1, 2, 4, 5).permute(0, 1, 2, 4, 3, 5)
    x = x.reshape(B, 1, LAT_W, LAT_C).squeeze(1)
    return x.permute(0, 2, 1)


def dit_forward(sd: SD, inp: Tensor, t: Tensor, text: Optional[Tensor],
                taps: Optional[dict] = None) -> Tensor:
    """Transformer.forward, transformer.py:158-193.  inp (B,64,30) -> (B,64,30)."""
    x = dit_patchify(sd, inp)
    c = time_embedding(t)
    if text is not None:
        c = c + text
    if taps is not None:
        taps["tokens"] = x
        taps["c"] = c
    for i in range(N_BLOCKS):
        x = dit_block(sd, i, x, c, taps)
    return dit_unpatchify(sd, x)


# --------------------------------------------------------------------------
# LA-VAE  (model/pretrained/vqvae.py)
# --------------------------------------------------------------------------
def _residual_stack(sd: SD, prefix: str, x: Tensor, n_layers: int) -> Tensor:
    """Residual / ResidualStack, vqvae.py:7-33.  The reference's first ReLU is
    in-place (nn.ReLU(True)), so the skip connection adds the RELU'd input:
    x <- relu(x) + block(relu(x)).  (x + self._block(x) evaluates x after the
    in-place op has mutated it.)"""
    for i in range(n_layers):
        p = f"{prefix}._layers.{i}._block."
        x = F.relu(x)
        h = F.conv1d(x, sd[p + "1.weight"], None, padding=1)
        h = F.relu(h)
        h = F.conv1d(h, sd[p + "3.weight"], None)
        x = x + h
    return F.relu(x)


def _count_res_layers(sd: SD, prefix: str) -> int:
    n = 0
    while f"{prefix}._layers.{n}._block.1.weight" in sd:
        n += 1
    return n


def vae_encode(sd: SD, x: Tensor, prefix: str = "encoder") -> Tuple[Tensor, Tensor]:
    """Encoder.forward, vqvae.py:57-71.  x (B,L) -> (z (B,64,30), before (B,64,L/4))."""
    p = prefix + "."
    h = x.view(x.shape[0], 1, x.shape[-1])
    h = F.relu(F.conv1d(h, sd[p + "_conv_1.weight"], sd[p + "_conv_1.bias"], stride=2, padding=1))
    h = F.relu(F.conv1d(h, sd[p + "_conv_2.weight"], sd[p + "_conv_2.bias"], stride=2, padding=1))
    h = F.conv1d(h, sd[p + "_conv_3.weight"], sd[p + "_conv_3.bias"], padding=1)
    h = _residual_stack(sd, p + "_residual_stack", h, _count_res_layers(sd, p + "_residual_stack"))
    before = F.conv1d(h, sd[p + "_pre_vq_conv.weight"], sd[p + "_pre_vq_conv.bias"])
    z = F.interpolate(before, size=LAT_W, mode="linear", align_corners=True)
    return z, before


def vae_decode(sd: SD, z: Tensor, length: int, prefix: str = "decoder") -> Tuple[Tensor, Tensor]:
    """Decoder.forward, vqvae.py:97-105.  z (B,64,30) -> (recon squeeze, after (B,64,L/4)).

    torch.squeeze drops every size-1 dim, so B==1 yields shape (L,)."""
    p = prefix + "."
    after = F.interpolate(z, size=int(length / 4), mode="linear", align_corners=True)
    h = F.conv1d(after, sd[p + "_conv_1.weight"], sd[p + "_conv_1.bias"], padding=1)
    h = _residual_stack(sd, p + "_residual_stack", h, _count_res_layers(sd, p + "_residual_stack"))
    h = F.relu(F.conv_transpose1d(h, sd[p + "_conv_trans_1.weight"], sd[p + "_conv_trans_1.bias"],
                                  stride=2, padding=1))
    h = F.conv_transpose1d(h, sd[p + "_conv_trans_2.weight"], sd[p + "_conv_trans_2.bias"],
                           stride=2, padding=1)
    return torch.squeeze(h), after


# --------------------------------------------------------------------------
# Sampling chains  (infer.py:73-95)
# --------------------------------------------------------------------------
def sample_ddpm(sd: SD, x_T: Tensor, text: Tensor, steps: int, cfg: float,
                noises, on_step=None) -> Tensor:
    """infer.py:75-88 (ddpm branch) with x_T and the per-step Gaussian draws
    injected.  ``noises`` is indexable by j (loop index, t = steps-1-j)."""
    tab = ddpm_tables(steps)
    x = x_T
    for j in range(steps):
        t = torch.full((x.size(0),), math.floor(steps - 1 - j), dtype=torch.long)
        u = dit_forward(sd, x, t, None)
        c = dit_forward(sd, x, t, text)
        pred = u + cfg * (c - u)
        x = ddpm_p_sample(tab, x, pred, t, noises[j])
        if on_step is not None:
            on_step(j, x)
    return x


def sample_rf(sd: SD, x_0: Tensor, text: Tensor, steps: int, cfg: float, on_step=None) -> Tensor:
    """infer.py:77-82 (flowmatching branch)."""
    x = x_0
    for j in range(steps):
        t = torch.round(torch.full((x.shape[0],), j * 1.0 / steps) * steps) / steps
        u = dit_forward(sd, x, t, None)
        c = dit_forward(sd, x, t, text)
        pred = u + cfg * (c - u)
        x = rf_euler(x, pred, 1.0 / steps)
        if on_step is not None:
            on_step(j, x)
    return x


# --------------------------------------------------------------------------
# MLP denoiser  (model/denoiser/mlp.py) -- config-1 plumbing
# --------------------------------------------------------------------------
def mlp_denoiser_forward(sd: SD, inp: Tensor, t: Tensor, text: Optional[Tensor]) -> Tensor:
    """MLP.forward / MLPlayer.forward, mlp.py:71-94.  inp (B,64,6)."""
    x = inp
    n_layers = 0
    while f"layers.{n_layers}.norm2.weight" in sd:
        n_layers += 1
    for i in range(n_layers):
        p = f"layers.{i}."
        te = time_embedding(t, 64).unsqueeze(-1)
        h = (x + te).permute(0, 2, 1)
        if text is not None:
            enc = text.unsqueeze(1).repeat(1, 6, 1)
            B, T, _ = h.shape
            nh, C = 4, 64
            k = F.linear(enc, sd[p + "cross_attn.key.weight"], sd[p + "cross_attn.key.bias"]
                         ).view(B, 6, nh, C // nh).transpose(1, 2)
            q = F.linear(h, sd[p + "cross_attn.query.weight"], sd[p + "cross_attn.query.bias"]
                         ).view(B, T, nh, C // nh).transpose(1, 2)
            v = F.linear(enc, sd[p + "cross_attn.value.weight"], sd[p + "cross_attn.value.bias"]
                         ).view(B, 6, nh, C // nh).transpose(1, 2)
            att = F.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(k.size(-1))), dim=-1)
            y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
            h = h + F.linear(y, sd[p + "cross_attn.proj.weight"], sd[p + "cross_attn.proj.bias"])
        h = F.layer_norm(h, (64,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        h = h + F.linear(F.relu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])),
                         sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
        h = h.permute(0, 2, 1)
        x = F.linear(F.relu(F.linear(h, sd[p + "mlp2.0.weight"], sd[p + "mlp2.0.bias"])),
                     sd[p + "mlp2.2.weight"], sd[p + "mlp2.2.bias"])
    return x


# --------------------------------------------------------------------------
# Device RNG specification (Philox4x32-10 + Box-Muller).  This is the BUILD's
# own perf-mode noise generator (the reference uses torch's CPU generator,
# DDPM.py:35 / infer.py:75, which no GPU stream can reproduce); restated here
# in numpy so tests can check the HIP generator bit-for-bit on the integer
# stage and to 1e-6 on the float stage.
# --------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr (...,4) uint32, key (...,2) uint32 -> (...,4) uint32."""
    c = ctr.astype(np.uint32).copy()
    k0 = key[..., 0].astype(np.uint32).copy()
    k1 = key[..., 1].astype(np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c[..., 0].astype(np.uint64)
            p1 = _PHILOX_M1 * c[..., 2].astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            n0 = hi1 ^ c[..., 1] ^ k0
            n1 = lo1
            n2 = hi0 ^ c[..., 3] ^ k1
            n3 = lo0
            c = np.stack([n0, n1, n2, n3], axis=-1)
            k0 = (k0 + _PHILOX_W0).astype(np.uint32)
            k1 = (k1 + _PHILOX_W1).astype(np.uint32)
    return c


def device_uniform(seed: int, stream: int, row0: int, n_rows: int, row_elems: int = 1) -> np.ndarray:
    """t2s_philox_uniform: element e of GLOBAL row r = lane e % 4 of ctr (e // 4, r, stream, 0), key = seed, as the
    24-bit uniform (x >> 8) * 2^-24 in [0, 1).  -> (n_rows, row_elems) float32."""
    q = (row_elems + 3) // 4
    ctr = np.zeros((n_rows, q, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(q, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.arange(row0, row0 + n_rows, dtype=np.uint32)[:, None]
    ctr[..., 2] = np.uint32(stream & 0xFFFFFFFF)
    key = np.zeros((n_rows, q, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    x = philox4x32_10(ctr, key).reshape(n_rows, q * 4)[:, :row_elems]
    return ((x >> np.uint32(8)).astype(np.float64) * 2.0 ** -24).astype(np.float32)


def device_normal(seed: int, stream: int, row0: int, n_rows: int, row_elems: int = LAT_C * LAT_W
                  ) -> np.ndarray:
    """The HIP generator's definition (t2ms_amd/csrc/t2s_sampler.hip, t2s_philox_normal):

    element e (0..row_elems) of GLOBAL row r draws from
      ctr = (e // 4, r, stream, 0), key = (seed_lo, seed_hi); lane = e % 4;
    lanes (0,1) and (2,3) are Box-Muller pairs:
      u1 = ((x_a >> 8) + 1) * 2^-24   in (0,1],  u2 = (x_b >> 8) * 2^-24  in [0,1)
      r = sqrt(-2 ln u1);  z_a = r cos(2 pi u2);  z_b = r sin(2 pi u2)
    ``stream`` is the sampling step index (x_T uses stream = 0xFFFFFFFF), so the
    draw is independent of how rows are sharded over GPUs.
    """
    assert row_elems % 4 == 0
    q = row_elems // 4
    rows = np.arange(row0, row0 + n_rows, dtype=np.uint32)
    ctr = np.zeros((n_rows, q, 4), dtype=np.uint32)
    ctr[..., 0] = np.arange(q, dtype=np.uint32)[None, :]
    ctr[..., 1] = rows[:, None]
    ctr[..., 2] = np.uint32(stream & 0xFFFFFFFF)
    key = np.zeros((n_rows, q, 2), dtype=np.uint32)
    key[..., 0] = np.uint32(seed & 0xFFFFFFFF)
    key[..., 1] = np.uint32((seed >> 32) & 0xFFFFFFFF)
    x = philox4x32_10(ctr, key)
    out = np.empty((n_rows, q, 4), dtype=np.float32)
    for a, b in ((0, 1), (2, 3)):
        u1 = ((x[..., a] >> np.uint32(8)).astype(np.float64) + 1.0) * 2.0 ** -24
        u2 = (x[..., b] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
        rad = np.sqrt(-2.0 * np.log(u1))
        out[..., a] = (rad * np.cos(2.0 * np.pi * u2)).astype(np.float32)
        out[..., b] = (rad * np.sin(2.0 * np.pi * u2)).astype(np.float32)
    return out.reshape(n_rows, row_elems)


# ---------------------------------------------------------------------------- evaluation.py:166-206
def eval_mse(ori, gen):
    """calculate_mse (evaluation.py:166-180): mean over samples of the per-sample, per-series mean squared error."""
    import numpy as np
    ori, gen = np.asarray(ori, dtype=np.float64), np.asarray(gen, dtype=np.float64)
    return float(np.mean([np.mean([np.mean((ori[i, :, j] - gen[i, :, j]) ** 2) for j in range(ori.shape[2])])
                          for i in range(ori.shape[0])]))


def eval_wape(ori, gen):
    """calculate_wape (evaluation.py:183-206): nanmean over samples of sum |ori - gen| / sum |ori| (NaN if 0)."""
    import numpy as np
    ori, gen = np.asarray(ori, dtype=np.float64), np.asarray(gen, dtype=np.float64)
    vals = []
    for i in range(ori.shape[0]):
        ae, av = np.abs(ori[i] - gen[i]).sum(), np.abs(ori[i]).sum()
        vals.append(ae / av if av != 0 else np.nan)
    return float(np.nanmean(np.asarray(vals)))



def eval_cosine(a, b):
    """cosine_similarity (Dataset_Construction_Pipeline/Evaluate_Datasets.py:6-15): flattened dot / norms, nan_to_num."""
    import numpy as np
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.sum(a * b) / (np.linalg.norm(a) * np.linalg.norm(b))
    return float(np.nan_to_num(s))


def eval_mrr(ori, gen, threshold=0.5):
    """calculate_mrr (evaluation.py:21-45) with k = all runs: ori (N, L, S), gen (N, L, S, G).  The runs are
    visited by descending similarity (reversed argsort) and the first one above the threshold gives
    rank = its run index + 1 (:37-41; the index, not the position in the ordering); 0 if none."""
    import numpy as np
    ori, gen = np.asarray(ori), np.asarray(gen)
    scores = np.zeros(ori.shape[0])
    for i in range(ori.shape[0]):
        sims = [eval_cosine(ori[i], gen[i, :, :, g]) for g in range(gen.shape[3])]
        rank = None
        for idx in np.argsort(sims, kind="stable")[::-1]:
            if sims[idx] > threshold:
                rank = idx + 1
                break
        scores[i] = 1.0 / rank if rank is not None else 0.0
    return float(np.mean(scores))


def eval_crps(ori, gen):
    """calculate_crps (evaluation.py:51-83): ori (N, L, S), gen (N, L, S, G).  Per sample / series / run: a Gaussian
    N(mean, std) is fitted to the generated series over TIME (population std, +1e-8 when 0), its CDF at the observed
    values is compared with the step function 1[obs >= mean], squared and averaged over time; then the mean over
    runs, series, samples."""
    import numpy as np
    from scipy.stats import norm
    ori, gen = np.asarray(ori), np.asarray(gen)
    vals = []
    for i in range(ori.shape[0]):
        total = 0.0
        for j in range(ori.shape[2]):
            per_run = []
            for k in range(gen.shape[3]):
                g = gen[i, :, j, k]
                mean, std = g.mean(), g.std()
                if std == 0:
                    std += 1e-8
                obs = ori[i, :, j]
                step = np.where(obs < mean, 0, 1)
                per_run.append(np.mean((step - norm.cdf(obs, loc=mean, scale=std)) ** 2))
            total += np.mean(per_run)
        vals.append(total / ori.shape[2])
    return float(np.asarray(vals).mean())


def eval_ed(ori, gen):
    """calculate_ed (evaluation.py:137-150): mean over samples of the per-series Euclidean distance over time."""
    import numpy as np
    ori, gen = np.asarray(ori), np.asarray(gen)
    d = np.linalg.norm(ori - gen, axis=1)          # (N, S)
    return float(d.mean(axis=1).mean())


def eval_dtw(ori, gen):
    """calculate_dtw (evaluation.py:152-163) = dtaidistance 2.3 dtw_ndim.distance(s1, s2) per sample, averaged:
    sqrt of the minimal-cost warping path with squared-Euclidean point costs between the (L, S) sequences, no window,
    no penalty (third-party, absent here: restated from its published definition -- UNPINNED)."""
    import numpy as np
    ori, gen = np.asarray(ori, dtype=np.float64), np.asarray(gen, dtype=np.float64)
    out = []
    for a, b in zip(ori, gen):
        n, m = a.shape[0], b.shape[0]
        cost = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        acc = np.full((n + 1, m + 1), np.inf)
        acc[0, 0] = 0.0
        for i in range(1, n + 1):
            for j in range(1, m + 1):
                acc[i, j] = cost[i - 1, j - 1] + min(acc[i - 1, j], acc[i, j - 1], acc[i - 1, j - 1])
        out.append(np.sqrt(acc[n, m]))
    return float(np.mean(out))


def eval_fid(act1, act2):
    """calculate_fid (evaluation.py:127-135): |mu1 - mu2|^2 + tr(S1 + S2 - 2 sqrtm(S1 S2)) (real part)."""
    import numpy as np
    from scipy.linalg import sqrtm
    act1, act2 = np.asarray(act1), np.asarray(act2)
    mu1, s1 = act1.mean(axis=0), np.cov(act1, rowvar=False)
    mu2, s2 = act2.mean(axis=0), np.cov(act2, rowvar=False)
    covmean = sqrtm(s1.dot(s2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum((mu1 - mu2) ** 2.0) + np.trace(s1 + s2 - 2.0 * covmean))


def ts2vec_encode(sd: SD, x: Tensor) -> Tuple[Tensor, Tensor]:
    """TSEncoder.forward in eval mode with mask 'all_true' (evaluate/ts2vec.py:366-399) and the 'full_series' pooling
    of TS2Vec.encode (:236-245): x (B, T, C_in) -> (per-step representation (B, T, C_out), max over time (B, C_out)).
    Time steps holding a NaN are zeroed before AND after input_fc (:367-368,388-389); the encoder is depth+1 ConvBlocks
    (:421-434: residual [1x1 projector when the widths differ or in the final block] + conv(gelu(conv(gelu(x)))), k=3,
    dilation 2^i, 'same' padding); dropout is the identity in eval mode."""
    x = x.clone()
    ok = ~x.isnan().any(dim=-1)
    x[~ok] = 0
    h = F.linear(x, sd["input_fc.weight"], sd["input_fc.bias"])
    h[~ok] = 0
    h = h.transpose(1, 2)
    i = 0
    while f"feature_extractor.net.{i}.conv1.conv.weight" in sd:
        p = f"feature_extractor.net.{i}."
        d = 2 ** i
        res = h
        if p + "projector.weight" in sd:
            res = F.conv1d(h, sd[p + "projector.weight"], sd[p + "projector.bias"])
        y = F.conv1d(F.gelu(h), sd[p + "conv1.conv.weight"], sd[p + "conv1.conv.bias"], padding=d, dilation=d)
        y = F.conv1d(F.gelu(y), sd[p + "conv2.conv.weight"], sd[p + "conv2.conv.bias"], padding=d, dilation=d)
        h = y + res
        i += 1
    rep = h.transpose(1, 2)
    return rep, rep.max(dim=1).values
