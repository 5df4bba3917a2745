"""Training-side pieces of the T2S path (train.py:101-131): loss and, later, the DiT backward."""
from __future__ import annotations

import torch

from . import _lib as L


def mse_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """F.mse_loss(a, b) (DDPM.py:37-38, rectified_flow.py:13-16) via t2s_mse (deterministic order)."""
    if not (a.is_cuda and b.is_cuda):
        raise L.T2SError("mse_loss: tensors must live on a GPU; the HIP path has no CPU fallback")
    if a.shape != b.shape:
        raise L.T2SError(f"mse_loss: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _MseFn.apply(a, b)
    return _mse_value(a, b)


def _mse_value(a, b):
    ac, bc = L.as_f32(a.detach()), L.as_f32(b.detach())
    out = torch.empty((), device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        L.check(L.lib().t2s_mse(L.dev_ptr(ac), L.dev_ptr(bc), out.data_ptr(), ac.numel(),
                                L.stream_ptr(a.device)), "t2s_mse")
    return out


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _mse_value(a, b)

    @staticmethod
    def backward(ctx, g):
        raise L.T2SError("mse backward: the training path (DiT backward kernels) is not built yet")


def dit_forward_autograd(model, input, t, text_input):
    raise L.T2SError(
        "Transformer.forward was called with autograd enabled, but the DiT backward kernels are not built "
        "yet (SURVEY.md 8f rank 1).  Wrap sampling in torch.no_grad() (as infer.py:65 does).")
