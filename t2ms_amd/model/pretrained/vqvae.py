"""Host-side mirror of the reference LA-VAE codec (model/pretrained/vqvae.py).

Class and attribute names equal the reference's, so its whole-module pickles
(``final_model.pth``, loaded with weights_only=False at infer.py:39 / train.py:22)
resolve to these classes and their tensors are used as-is.  ``Encoder.forward``
and ``Decoder.forward`` run single-launch HIP kernels (t2s_vae.hip); the
nn.Conv1d objects only hold weights.  Inference only (the reference freezes
the VAE while training the DiT, train.py:31-33); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
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
        self._fin = weakref.finalize(self, L.lib().t2s_vae_destroy, self.ptr)

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
        stamp = (str(device),) + tuple((t.data_ptr(), t._version) for t in ts)
        h = self.__dict__.get("_t2s_h")
        if h is None or self.__dict__.get("_t2s_stamp") != stamp:
            if h is not None:
                h.close()
            w, keep = self._weights_struct()
            h = _VaeHandle(w, device)
            del keep  # the library made its own copies
            self.__dict__["_t2s_h"], self.__dict__["_t2s_stamp"] = h, stamp
        return h.ptr

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_t2s_h", None)
        state.pop("_t2s_stamp", None)
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

    def _forward_autograd(self, inputs):
        """The same forward as torch ops UNDER AUTOGRAD, for the one case that needs the encoder's gradients: train.py:31-33 with
        `usepretrainedvae` false (the LA-VAE encoder trained jointly with the denoiser).  Host-level plumbing like the MLP
        denoiser: no scripted configuration of the reference trains the encoder; the frozen / inference path is the HIP kernel
        below.  nn.ReLU(True) of the reference's Residual mutates the block input, so the skip carries relu(x) (vqvae.py:10-21)."""
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
            return self._forward_autograd(inputs)
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
