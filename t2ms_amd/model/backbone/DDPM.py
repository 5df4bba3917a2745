"""Host-side mirror of the reference DDPM backbone (model/backbone/DDPM.py:10-38).

Same constructor, attributes (beta, alpha, alpha_bar, sigma2, total_steps) and
methods; the per-element arithmetic runs in libt2s_hip.so (t2s_sampler.hip).
The schedule tables are built ONCE on the host with the reference's own fp32
torch ops (linspace / cumprod), so every GPU sees bit-identical coefficients.
"""
from __future__ import annotations

from typing import Optional

import torch

from ... import _lib as L


def ddpm_host_tables(total_steps: int):
    """DDPM.__init__ (DDPM.py:14-18) on the host + the derived per-step coefficients the
    kernels read: coef[t] = {1/sqrt(alpha), (1-alpha)/sqrt(1-alpha_bar), sqrt(beta)} (DDPM.py:30-36)
    and {sqrt(alpha_bar), sqrt(1-alpha_bar)} (DDPM.py:20-21,27)."""
    beta = torch.linspace(0.0001, 0.02, total_steps)
    alpha = 1 - beta
    alpha_bar = torch.cumprod(alpha, dim=0)
    coef = torch.stack([1 / (alpha ** 0.5), (1 - alpha) / (1 - alpha_bar) ** .5, beta ** .5], dim=1).contiguous()
    return dict(beta=beta, alpha=alpha, alpha_bar=alpha_bar, coef=coef,
                sqrt_ab=(alpha_bar ** 0.5).contiguous(), sqrt_1mab=((1 - alpha_bar) ** 0.5).contiguous())


def gather(consts: torch.Tensor, t: torch.Tensor):
    """DDPM.py:7-9 (index helper kept for API compatibility)."""
    return consts.gather(-1, t).reshape(-1, 1, 1)


class DDPM:
    def __init__(self, total_steps: int, device):
        self.device = device
        self.total_steps = total_steps
        tab = ddpm_host_tables(total_steps)
        self.beta = tab["beta"].to(device)
        self.alpha = tab["alpha"].to(device)
        self.alpha_bar = tab["alpha_bar"].to(device)
        self.sigma2 = self.beta
        self._coef = tab["coef"].to(device)
        self._sqrt_ab = tab["sqrt_ab"].to(device)
        self._sqrt_1mab = tab["sqrt_1mab"].to(device)

    def _rows(self, x: torch.Tensor, t: torch.Tensor, **same_shape):
        """Checks before raw device pointers reach the kernels: x (B,C,W) on the GPU, t (B,), and every tensor in
        `same_shape` shaped like x (a (B,64,30) DiT prediction against a (B,64,6) MLP latent would otherwise be read
        out of bounds).  The VALUES of t are checked on the device (a row outside [0,T) comes out NaN)."""
        for name, v in same_shape.items():
            if v is not None and tuple(v.shape) != tuple(x.shape):
                raise L.T2SError(f"DDPM: {name} must have the latent's shape {tuple(x.shape)}, got {tuple(v.shape)}")
        if not x.is_cuda:
            raise L.T2SError("DDPM: tensors must live on a GPU; the HIP path has no CPU fallback")
        if x.dim() != 3 or (x.shape[1] * x.shape[2]) % 4 != 0:
            raise L.T2SError(f"DDPM: latent must be (B,C,W) with C*W a multiple of 4 ((B,64,30) on the DiT path), "
                             f"got {tuple(x.shape)}")
        if not t.is_cuda and t.numel() and (int(t.min()) < 0 or int(t.max()) >= self.total_steps):
            # a host-side t costs nothing to check; the reference's gather (DDPM.py:7-9) raises for it.  (A device-side t
            # is checked by the kernels: a row outside the table comes out NaN rather than read out of bounds.)
            raise IndexError(f"DDPM: t must lie in [0, {self.total_steps}), got [{int(t.min())}, {int(t.max())}]")
        ti = t.to(device=x.device, dtype=torch.int32).contiguous()
        if ti.shape != (x.shape[0],):
            raise L.T2SError(f"DDPM: t must be ({x.shape[0]},), got {tuple(ti.shape)}")
        return ti

    def q_xt_x0(self, x0: torch.Tensor, t: torch.Tensor):
        """DDPM.py:19-22 (mean, var) -- small helper, evaluated through q_sample's kernel with eps=0/1."""
        zeros = torch.zeros_like(x0)
        mean, _ = self.q_sample(x0, t, zeros)
        var = (1 - gather(self.alpha_bar, t.long()))
        return mean, var.to(self.device)

    def q_sample(self, x0: torch.Tensor, t: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """DDPM.py:23-27: x_t = sqrt(ab_t) x0 + sqrt(1-ab_t) eps."""
        ti = self._rows(x0, t, eps=eps)
        if eps is None:
            eps = torch.randn_like(x0)
        x0c, epsc = L.as_f32(x0), L.as_f32(eps)
        out = torch.empty_like(x0c)
        with torch.cuda.device(x0.device):
            L.check(L.lib().t2s_ddpm_q_sample_n(L.dev_ptr(x0c, "x0"), L.dev_ptr(epsc, "eps"),
                                                L.dev_ptr(ti, "t", torch.int32), L.dev_ptr(self._sqrt_ab),
                                                L.dev_ptr(self._sqrt_1mab), L.dev_ptr(out), x0.shape[0],
                                                x0.shape[1] * x0.shape[2], int(self.total_steps),
                                                L.stream_ptr(x0.device)), "t2s_ddpm_q_sample")
        return out, eps

    def p_sample(self, xt: torch.Tensor, n_xt: torch.Tensor, t: torch.Tensor, eps: Optional[torch.Tensor] = None):
        """DDPM.py:28-36.  ``eps`` (optional, extension) injects the Gaussian draw; by default it is
        drawn with torch.randn on xt's device like the reference (noise is added at t=0 too)."""
        ti = self._rows(xt, t, n_xt=n_xt, eps=eps)
        if eps is None:
            eps = torch.randn(xt.shape, device=xt.device)
        x, e, z = L.as_f32(xt), L.as_f32(n_xt), L.as_f32(eps)
        out = torch.empty_like(x)
        with torch.cuda.device(xt.device):
            L.check(L.lib().t2s_ddpm_p_sample_n(L.dev_ptr(x, "xt"), L.dev_ptr(e, "n_xt"),
                                                L.dev_ptr(ti, "t", torch.int32), L.dev_ptr(z, "eps"),
                                                L.dev_ptr(self._coef), L.dev_ptr(out), xt.shape[0],
                                                xt.shape[1] * xt.shape[2], int(self.total_steps),
                                                L.stream_ptr(xt.device)), "t2s_ddpm_p_sample")
        return out

    def loss(self, n_gt: torch.Tensor, n_xt: torch.Tensor):
        """DDPM.py:37-38 (F.mse_loss)."""
        from ...train import mse_loss
        return mse_loss(n_gt, n_xt)


DDPM.__module__ = "model.backbone.DDPM"
