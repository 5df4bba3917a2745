"""Host-side mirror of the reference LA-VAE codec (model/pretrained/vqvae.py).

Class and attribute names equal the reference's, so its whole-module pickles
(``final_model.pth``, loaded with weights_only=False at infer.py:39 / train.py:22)
resolve to these classes and their tensors are used as-is.  ``Encoder.forward``
and ``Decoder.forward`` run single-launch HIP kernels (t2s_vae.hip); the
nn.Conv1d objects only hold weights.  The reference freezes the VAE while training
the DiT (train.py:31-33); with `usepretrainedvae` false the encoder trains and its
backward is t2s_vae_encode_backward.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
import torch.nn as nn

from ... import _lib as L
from .core import BaseModel


class Residual(nn.Module):
    """vqvae.py:7-22 (parameter container: _block = [ReLU, Conv k3 no-bias, ReLU, Conv k1 no-bias])."""

    def __init__(self, in_channels, num_hiddens, num_residual_hiddens):
        super().__init__()
        self._block = nn.Sequential(
            nn.ReLU(True),
            nn.Conv1d(in_channels, num_residual_hiddens, kernel_size=3, stride=1, padding=1, bias=False),
            nn.ReLU(True),
            nn.Conv1d(num_residual_hiddens, num_hiddens, kernel_size=1, stride=1, bias=False))

    def forward(self, x):
        raise L.T2SError("Residual runs inside the fused LA-VAE kernels; call Encoder/Decoder.forward")


class ResidualStack(nn.Module):
    """vqvae.py:24-33."""

    def __init__(self, in_channels, num_hiddens, num_residual_layers, num_residual_hiddens):
        super().__init__()
        self._num_residual_layers = num_residual_layers
        self._layers = nn.ModuleList([Residual(in_channels, num_hiddens, num_residual_hiddens)
                                      for _ in range(num_residual_layers)])

    def forward(self, x):
        raise L.T2SError("ResidualStack runs inside the fused LA-VAE kernels; call Encoder/Decoder.forward")


class _VaeHandle:
    def __init__(self, w: L.VaeWeights, device):
        self.ptr = C.c_void_p()
        torch.cuda.synchronize(device)
        with torch.cuda.device(device):
            L.check(L.lib().t2s_vae_create(C.byref(w), C.byref(self.ptr)), "t2s_vae_create")
        self._fin = weakref.finalize(self, L.destroy_locked, "t2s_vae_destroy", str(torch.device(device)), self.ptr)

    def close(self):
        self._fin()


def _stack_ptrs(stack: ResidualStack, dst: L.VaeStackWeights, keep):
    n = len(stack._layers)
    if n > 4:
        raise L.T2SError(f"LA-VAE: num_residual_layers={n} > 4 is not supported by the HIP codec")
    for i, layer in enumerate(stack._layers):
        c3 = L.as_f32(layer._block[1].weight.detach())
        c1 = L.as_f32(layer._block[3].weight.detach())
        keep += [c3, c1]
        dst.conv3_w[i], dst.conv1_w[i] = c3.data_ptr(), c1.data_ptr()
    return n


class _Codec(nn.Module):
    """Shared handle cache for Encoder / Decoder (keyed on parameter storage + version)."""

    def _tensors(self):
        return [p for p in self.parameters()]

    def _handle(self, device):
        device = torch.device(device)
        ts = self._tensors()
        for t in ts:
            if t.device != device:
                raise L.T2SError(f"LA-VAE parameters live on {t.device} but the input is on {device}")
        inference = any(t.is_inference() for t in ts)          # no version counters: nothing to cache against
        stamp = None if inference else (str(device),) + tuple((t.data_ptr(), t._version) for t in ts)
        shape = (str(device),) + tuple(tuple(t.shape) for t in ts)
        h = self.__dict__.get("_t2s_h")
        if h is None or self.__dict__.get("_t2s_shape") != shape:
            with L.device_lock(device):       # frees, allocations, a stream synchronisation: not inside another thread's capture
                if h is not None:
                    h.close()
                w, keep = self._weights_struct()
                h = _VaeHandle(w, device)
                del keep  # the library made its own copies
            self.__dict__["_t2s_h"], self.__dict__["_t2s_stamp"], self.__dict__["_t2s_shape"] = h, stamp, shape
        elif stamp is None or self.__dict__.get("_t2s_stamp") != stamp:
            # same tensors' shapes, new contents (an optimizer step on a trainable encoder, load_state_dict): re-copy into the
            # handle's own buffers -- no allocation, stream-ordered (t2s_vae_update_weights)
            w, keep = self._weights_struct()
            with torch.cuda.device(device):
                L.check(L.lib().t2s_vae_update_weights(h.ptr, C.byref(w), L.stream_ptr(device)), "t2s_vae_update_weights")
            h.keep = keep          # the copies are stream-ordered: keep the sources alive until the next refresh
            self.__dict__["_t2s_stamp"] = stamp
        return h.ptr

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_t2s_h", None)
        state.pop("_t2s_stamp", None)
        state.pop("_t2s_shape", None)
        return state


class Encoder(_Codec):
    """vqvae.py:36-71."""

    def __init__(self, in_channels, num_hiddens, num_residual_layers, num_residual_hiddens, embedding_dim):
        super().__init__()
        self._conv_1 = nn.Conv1d(in_channels, num_hiddens // 2, kernel_size=4, stride=2, padding=1)
        self._conv_2 = nn.Conv1d(num_hiddens // 2, num_hiddens, kernel_size=4, stride=2, padding=1)
        self._conv_3 = nn.Conv1d(num_hiddens, num_hiddens, kernel_size=3, stride=1, padding=1)
        self._residual_stack = ResidualStack(num_hiddens, num_hiddens, num_residual_layers, num_residual_hiddens)
        self._pre_vq_conv = nn.Conv1d(num_hiddens, embedding_dim, kernel_size=1, stride=1)

    def _weights_struct(self):
        w, keep = L.VaeWeights(), []
        w.hidden = self._conv_2.out_channels
        w.emb = self._pre_vq_conv.out_channels
        w.res_hidden = self._residual_stack._layers[0]._block[1].out_channels if len(self._residual_stack._layers) else 1
        w.n_res_layers = _stack_ptrs(self._residual_stack, w.enc_stack, keep)
        for name, t in (("enc_conv1_w", self._conv_1.weight), ("enc_conv1_b", self._conv_1.bias),
                        ("enc_conv2_w", self._conv_2.weight), ("enc_conv2_b", self._conv_2.bias),
                        ("enc_conv3_w", self._conv_3.weight), ("enc_conv3_b", self._conv_3.bias),
                        ("enc_prevq_w", self._pre_vq_conv.weight), ("enc_prevq_b", self._pre_vq_conv.bias)):
            c = L.as_f32(t.detach())
            keep.append(c)
            setattr(w, name, c.data_ptr())
        return w, keep

    def _hip_backward_ok(self, Ln):
        """t2s_vae_encode_backward covers the reference's default LA-VAE (pretrained_lavae_unified.py:119-122: hidden 128,
        res_hidden 128 / 256, emb 64) on the BASELINE lengths (L <= 128)."""
        rh = self._residual_stack._layers[0]._block[1].out_channels if len(self._residual_stack._layers) else 128
        return (self._conv_2.out_channels == 128 and rh % 128 == 0 and self._pre_vq_conv.out_channels == 64
                and len(self._residual_stack._layers) <= 4 and 8 <= Ln <= 128 and Ln % 4 == 0)

    def _grad_params(self):
        """The 12 encoder tensors in t2s_vae_enc_grads order."""
        ps = [self._conv_1.weight, self._conv_1.bias, self._conv_2.weight, self._conv_2.bias, self._conv_3.weight, self._conv_3.bias]
        ps += [layer._block[1].weight for layer in self._residual_stack._layers]
        ps += [layer._block[3].weight for layer in self._residual_stack._layers]
        return ps + [self._pre_vq_conv.weight, self._pre_vq_conv.bias]

    def _forward_autograd(self, inputs):
        """The same forward as torch ops UNDER AUTOGRAD: only for LA-VAE shapes t2s_vae_encode_backward does not cover
        (non-default hyper-parameters, L > 128) when the encoder trains (train.py:31-33 with `usepretrainedvae` false) --
        host-level plumbing like the MLP denoiser.  The default shape runs forward AND backward in the HIP kernels (_EncodeFn).
        nn.ReLU(True) of the reference's Residual mutates the block input, so the skip carries relu(x) (vqvae.py:10-21)."""
        import torch.nn.functional as F
        B, Ln = inputs.shape[0], inputs.shape[-1]
        h = inputs.float().reshape(B, 1, Ln)
        h = F.relu(self._conv_1(h))
        h = F.relu(self._conv_2(h))
        h = self._conv_3(h)
        for layer in self._residual_stack._layers:
            h = F.relu(h)
            h = h + layer._block[3](F.relu(layer._block[1](h)))
        h = F.relu(h)
        before = self._pre_vq_conv(h)
        return F.interpolate(before, size=L.LAT_W, mode="linear", align_corners=True), before

    def forward(self, inputs):
        """x (B,L) [or (B,1,L)] -> (z (B,64,30), before (B,64,L/4)); vqvae.py:57-71."""
        if not inputs.is_cuda:
            raise L.T2SError("Encoder.forward: input must live on a GPU; the HIP path has no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # (a series that itself requires a gradient -- never the case in train.py:104-106, where it is data -- takes the
            # torch-op forward: t2s_vae_encode_backward produces parameter gradients only)
            if (self._hip_backward_ok(inputs.shape[-1]) and not inputs.requires_grad
                    and os.environ.get("T2S_ENCODER_TORCH_AUTOGRAD", "0") in ("", "0")):
                return _EncodeFn.apply(self, inputs, *self._grad_params())
            return self._forward_autograd(inputs)
        return self._forward_hip(inputs)

    def _forward_hip(self, inputs):
        B, Ln = inputs.shape[0], inputs.shape[-1]
        x = L.as_f32(inputs).reshape(B, Ln)
        dev = x.device
        with torch.cuda.device(dev):
            h = self._handle(dev)
            emb = self._pre_vq_conv.out_channels
            z = torch.empty(B, emb, L.LAT_W, device=dev, dtype=torch.float32)
            before = torch.empty(B, emb, Ln // 4, device=dev, dtype=torch.float32)
            L.check(L.lib().t2s_vae_encode(h, L.dev_ptr(x, "inputs"), L.dev_ptr(z), L.dev_ptr(before), B, Ln,
                                           L.stream_ptr(dev)), "t2s_vae_encode")
        return z, before


class _EncodeFn(torch.autograd.Function):
    """Encoder.forward under autograd, both directions in the HIP kernels: t2s_vae_encode now, t2s_vae_encode_backward for the
    12 parameter gradients (the forward is recomputed from x there; nothing but x is saved).  The input series gets no
    gradient (it is data, train.py:104-106)."""

    @staticmethod
    def forward(ctx, enc, inputs, *params):
        with torch.no_grad():
            z, before = enc._forward_hip(inputs)
        ctx.enc = enc
        ctx.save_for_backward(L.as_f32(inputs).reshape(inputs.shape[0], inputs.shape[-1]))
        ctx.n_layers = len(enc._residual_stack._layers)
        ctx.set_materialize_grads(False)      # train.py uses z only: `before` then arrives as None, not as a zero tensor
        return z, before

    @staticmethod
    def backward(ctx, dz, dbefore):
        (x,) = ctx.saved_tensors
        enc, n = ctx.enc, ctx.n_layers
        dev = x.device
        params = enc._grad_params()
        grads = [torch.empty_like(p, dtype=torch.float32, memory_format=torch.contiguous_format) for p in params]
        g = L.VaeEncGrads()
        names = ["conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b"]
        for nm, t in zip(names, grads[:6]):
            setattr(g, nm, t.data_ptr())
        for i in range(n):
            g.stack_conv3_w[i] = grads[6 + i].data_ptr()
            g.stack_conv1_w[i] = grads[6 + n + i].data_ptr()
        g.prevq_w, g.prevq_b = grads[6 + 2 * n].data_ptr(), grads[7 + 2 * n].data_ptr()
        dzc = L.as_f32(dz) if dz is not None else torch.zeros(x.shape[0], 64, L.LAT_W, device=dev)
        dbc = L.as_f32(dbefore) if dbefore is not None else None
        with torch.cuda.device(dev):
            h = enc._handle(dev)           # (the weights of the forward: no optimizer step happens between the two)
            L.check(L.lib().t2s_vae_encode_backward(h, L.dev_ptr(x), L.dev_ptr(dzc), L.dev_ptr(dbc), C.byref(g), x.shape[0],
                                                    x.shape[1], L.stream_ptr(dev)), "t2s_vae_encode_backward")
        out = [gr if p.requires_grad else None for gr, p in zip(grads, params)]
        return (None, None, *out)


class Decoder(_Codec):
    """vqvae.py:74-105."""

    def __init__(self, in_channels, num_hiddens, num_residual_layers, num_residual_hiddens):
        super().__init__()
        self._conv_1 = nn.Conv1d(in_channels, num_hiddens, kernel_size=3, stride=1, padding=1)
        self._residual_stack = ResidualStack(num_hiddens, num_hiddens, num_residual_layers, num_residual_hiddens)
        self._conv_trans_1 = nn.ConvTranspose1d(num_hiddens, num_hiddens // 2, kernel_size=4, stride=2, padding=1)
        self._conv_trans_2 = nn.ConvTranspose1d(num_hiddens // 2, 1, kernel_size=4, stride=2, padding=1)

    def _weights_struct(self):
        w, keep = L.VaeWeights(), []
        w.hidden = self._conv_1.out_channels
        w.emb = self._conv_1.in_channels
        w.res_hidden = self._residual_stack._layers[0]._block[1].out_channels if len(self._residual_stack._layers) else 1
        w.n_res_layers = _stack_ptrs(self._residual_stack, w.dec_stack, keep)
        for name, t in (("dec_conv1_w", self._conv_1.weight), ("dec_conv1_b", self._conv_1.bias),
                        ("dec_ct1_w", self._conv_trans_1.weight), ("dec_ct1_b", self._conv_trans_1.bias),
                        ("dec_ct2_w", self._conv_trans_2.weight), ("dec_ct2_b", self._conv_trans_2.bias)):
            c = L.as_f32(t.detach())
            keep.append(c)
            setattr(w, name, c.data_ptr())
        return w, keep

    def forward(self, inputs, length):
        """z (B,64,W) -> (recon, after (B,64,L/4)); vqvae.py:97-105 (W = 30 on the DiT path; any W <= 32, e.g. the L/4
        of the MLP-denoiser path, as F.interpolate accepts).  ``recon`` follows torch.squeeze's shape rule: (B,L), or
        (L,) when B == 1."""
        if not inputs.is_cuda:
            raise L.T2SError("Decoder.forward: input must live on a GPU; the HIP path has no CPU fallback")
        z = L.as_f32(inputs)
        if z.dim() != 3 or not 1 <= z.shape[2] <= 32:
            raise L.T2SError(f"Decoder.forward: latent must be (B,C,W) with W <= 32, got {tuple(z.shape)}")
        B, dev = z.shape[0], z.device
        Ln = int(length / 4) * 4
        with torch.cuda.device(dev):
            h = self._handle(dev)
            recon = torch.empty(B, Ln, device=dev, dtype=torch.float32)
            after = torch.empty(B, z.shape[1], Ln // 4, device=dev, dtype=torch.float32)
            L.check(L.lib().t2s_vae_decode_w(h, L.dev_ptr(z, "inputs"), L.dev_ptr(recon), L.dev_ptr(after), B, Ln,
                                             z.shape[2], L.stream_ptr(dev)), "t2s_vae_decode")
        return torch.squeeze(recon.unsqueeze(1)), after


class vqvae(BaseModel):
    """vqvae.py:108-142 (no vector quantiser despite the name)."""

    def __init__(self, args):
        super().__init__()
        self.encoder = Encoder(1, args.block_hidden_size, args.num_residual_layers, args.res_hidden_size,
                               args.embedding_dim)
        self.decoder = Decoder(args.embedding_dim, args.block_hidden_size, args.num_residual_layers,
                               args.res_hidden_size)

    def shared_eval(self, batch, optimizer, mode):  # pyright: ignore[reportIncompatibleMethodOverride]
        """vqvae.py:118-135.  Only the val/test branch exists here: LA-VAE pre-training
        (pretrained_lavae_unified.py) is outside the accelerated path."""
        if mode == "train":
            raise L.T2SError("LA-VAE training is out of scope of the HIP path (frozen codec, train.py:31-33)")
        from ...train import mse_loss
        with torch.no_grad():
            z, before = self.encoder(batch)
            data_recon, after = self.decoder(z, length=batch.shape[-1])
            recon_error = mse_loss(data_recon.reshape(batch.shape), batch)
            loss = recon_error + mse_loss(before, after)
        return loss, recon_error, data_recon, z

    def forward(self, x):
        z, _ = self.encoder(x)
        return self.decoder(z, x.shape[-1])


for _cls in (Residual, ResidualStack, Encoder, Decoder, vqvae):
    _cls.__module__ = "model.pretrained.vqvae"
