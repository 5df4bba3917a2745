"""Host-side mirror of the reference rectified-flow backbone
(model/backbone/rectified_flow.py:4-16); arithmetic in libt2s_hip.so."""
from __future__ import annotations

import torch

from ... import _lib as L


class RectifiedFlow:
    def euler(self, x_t, v, dt):
        """rectified_flow.py:5-7: x_t + v*dt (out of place, like the reference's rebinding)."""
        if not x_t.is_cuda:
            raise L.T2SError("RectifiedFlow.euler: tensors must live on a GPU; no CPU fallback")
        out = L.as_f32(x_t).clone()
        vv = L.as_f32(v)
        if out.numel() % L.LAT != 0 or vv.shape != out.shape:
            raise L.T2SError(f"RectifiedFlow.euler: expected matching (B,64,30) tensors, got {tuple(x_t.shape)}, {tuple(v.shape)}")
        with torch.cuda.device(out.device):
            L.check(L.lib().t2s_rf_step(L.dev_ptr(out), L.dev_ptr(vv, "v"), None, 0.0, float(dt),
                                        out.numel() // L.LAT, L.stream_ptr(out.device)), "t2s_rf_step")
        return out

    def create_flow(self, x_1, t, x_0=None):
        """rectified_flow.py:8-12: x_t = t x_1 + (1-t) x_0, x_0 ~ N(0,1).  ``x_0`` (optional,
        extension) injects the draw."""
        if not x_1.is_cuda:
            raise L.T2SError("RectifiedFlow.create_flow: tensors must live on a GPU; no CPU fallback")
        if x_0 is None:
            x_0 = torch.randn_like(x_1)
        a, b = L.as_f32(x_1), L.as_f32(x_0)
        tf = L.as_f32(t.to(x_1.device))
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            L.check(L.lib().t2s_rf_create_flow(L.dev_ptr(a, "x_1"), L.dev_ptr(b, "x_0"), L.dev_ptr(tf, "t"),
                                               L.dev_ptr(out), a.shape[0], L.stream_ptr(a.device)),
                    "t2s_rf_create_flow")
        return out, x_0

    def loss(self, v, noise_gt):
        """rectified_flow.py:13-16 (F.mse_loss)."""
        from ...train import mse_loss
        return mse_loss(v, noise_gt)


RectifiedFlow.__module__ = "model.backbone.rectified_flow"
