"""Host-side mirror of the reference DiT denoiser (model/denoiser/transformer.py).

Same class name, constructor, parameter names/shapes/initialisation and
``forward(input=, t=, text_input=)`` contract as the reference, so
checkpoints and drivers are drop-in -- but ``forward`` runs the hand-written
HIP kernels of libt2s_hip.so (t2s_dit_forward).  The nn.Module tree below only
*holds parameters* under the reference's state-dict keys; no torch op computes
the network, and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import math
import weakref

import torch
import torch.nn as nn

from ... import _lib as L

__all__ = ["Transformer", "Transformerlayer", "TimeEmbedding", "modulate",
           "get_sinusoidal_positional_embeddings", "InverseLatentEmbedding", "LatentEmbedding"]

EMB, HEADS, DEPTH, PATCH = 128, 4, 4, 2
LAT_H, LAT_W = 30, 64           # the reference's self.H / self.W (transformer.py:132-133)
N_PATCH = (LAT_H // PATCH) * (LAT_W // PATCH)


def modulate(x, shift, scale):
    """transformer.py:7-8 -- kept for API compatibility; the HIP path fuses it into the GEMM prologue."""
    raise L.T2SError("modulate() is fused into the HIP kernels; call Transformer.forward instead")


def get_sinusoidal_positional_embeddings(num_positions, d_model):
    """Fixed sin/cos table (transformer.py:14-23): built once on the host at construction."""
    pos = torch.arange(num_positions).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2) * -(math.log(10000.0) / d_model)).unsqueeze(0)
    table = torch.zeros(num_positions, d_model)
    table[:, 0::2] = torch.sin(pos * div)
    table[:, 1::2] = torch.cos(pos * div)
    return table.unsqueeze(0)


class TimeEmbedding(nn.Module):
    """transformer.py:25-40.  Parameter-free; evaluated by t2s_time_embedding on the GPU."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0, "Dimension must be even"
        self.dim = dim

    def forward(self, t):
        if self.dim != EMB:
            raise L.T2SError("HIP TimeEmbedding is built for dim=128")
        tf = L.as_f32(t)
        if not tf.is_cuda:
            raise L.T2SError("TimeEmbedding: t must live on a GPU (no CPU fallback)")
        freqs = _freqs_on(tf.device)
        out = torch.empty(tf.shape[0], EMB, device=tf.device, dtype=torch.float32)
        with torch.cuda.device(tf.device):
            L.check(L.lib().t2s_time_embedding_freqs(L.dev_ptr(freqs), L.dev_ptr(tf, "t"), L.dev_ptr(out), tf.shape[0],
                                                     L.stream_ptr(tf.device)), "t2s_time_embedding_freqs")
        return out


class _Attention(nn.Module):
    """Parameter container with timm's Attention sub-module names (qkv, proj)."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, *_a, **_k):
        raise L.T2SError("attention runs inside t2s_dit_forward (t2s_attn.hip); call Transformer.forward")


class _Mlp(nn.Module):
    """Parameter container with timm's Mlp sub-module names (fc1, fc2)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, *_a, **_k):
        raise L.T2SError("the MLP runs inside t2s_dit_forward (t2s_gemm.h); call Transformer.forward")


class Transformerlayer(nn.Module):
    """adaLN-Zero DiT block parameters (transformer.py:94-124)."""

    def __init__(self):
        super().__init__()
        self.norm1 = nn.LayerNorm(EMB, elementwise_affine=False, eps=1e-6)
        self.norm2 = nn.LayerNorm(EMB, elementwise_affine=False, eps=1e-6)
        self.attn = _Attention(EMB, HEADS)
        self.mlp = _Mlp(EMB, int(EMB * 2.0))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(EMB, 6 * EMB, bias=True))

    def forward(self, *_a, **_k):
        raise L.T2SError("DiT blocks run inside t2s_dit_forward; call Transformer.forward")


class LatentEmbedding(nn.Module):
    """Dead code in the reference (transformer.py:46-62, never instantiated); kept importable."""

    def __init__(self, embed_dim: int = 64):
        super().__init__()
        self.dim = embed_dim
        self.embedding2d = nn.Conv2d(1, embed_dim, kernel_size=(6, 6), stride=(6, 6))


class InverseLatentEmbedding(nn.Module):
    """Instantiated as ``unpatch`` but never called by the reference forward
    (transformer.py:65-87,150): its tensors must exist in the state-dict, nothing else."""

    def __init__(self, embed_dim: int = 64):
        super().__init__()
        self.dim = embed_dim
        self.inv_embedding2d = nn.ConvTranspose2d(embed_dim, 1, kernel_size=(6, 6), stride=(6, 6))
        self.fc1 = nn.Linear(60, 128)
        self.fc2 = nn.Linear(128, 64)


# ---------------------------------------------------------------------------- device-side state
_FREQS = {}


def _freqs_on(device) -> torch.Tensor:
    """10000**linspace(0,1,64) evaluated with the reference's own fp32 torch ops on the host
    (transformer.py:34), then uploaded -- bit-identical table, device sin/cos."""
    key = str(device)
    if key not in _FREQS:
        _FREQS[key] = torch.pow(10000, torch.linspace(0, 1, EMB // 2)).to(device)
    return _FREQS[key]


_HANDLE_IDS = itertools.count(1)


class _DitHandle:
    """Owns one t2s_dit (packed weights + workspace) for one module on one device.  `uid` identifies THIS handle
    for the life of the process: a re-created handle often gets the freed one's address back, so users that cache
    state bound to a handle (Sampler's captured hipGraph, a pending backward) compare uids, never pointers."""

    def __init__(self, weights: L.DitWeights, keep, max_seqs: int):
        self.uid = next(_HANDLE_IDS)
        self.ptr = C.c_void_p()
        self.keep = keep
        self.max_seqs = max_seqs
        L.check(L.lib().t2s_dit_create(C.byref(weights), max_seqs, C.byref(self.ptr)), "t2s_dit_create")
        dev = f"cuda:{torch.cuda.current_device()}"          # created under torch.cuda.device(device)
        self._fin = weakref.finalize(self, L.destroy_locked, "t2s_dit_destroy", dev, self.ptr)

    def close(self):
        self._fin()


class Transformer(nn.Module):
    """Drop-in for ``model.denoiser.transformer.Transformer`` (transformer.py:128-204)."""

    def __init__(self):
        super().__init__()
        self.channel, self.H, self.W, self.patch_size = 1, LAT_H, LAT_W, PATCH
        self.patch_count = N_PATCH
        self.conv = nn.Conv2d(1, PATCH * PATCH, kernel_size=PATCH, padding=0, stride=PATCH)
        self.patch_emb = nn.Linear(PATCH * PATCH, EMB)
        self.pos_embed = nn.Parameter(get_sinusoidal_positional_embeddings(N_PATCH, EMB), requires_grad=False)
        self.ln = nn.LayerNorm(EMB)
        self.linear_emb_to_patch = nn.Linear(EMB, PATCH * PATCH)
        self.time_emb = TimeEmbedding(dim=EMB)
        self.layers = nn.ModuleList([Transformerlayer() for _ in range(DEPTH)])
        self.unpatch = InverseLatentEmbedding(embed_dim=EMB)
        self.initialize_weights()

    # -- a15: xavier on every Linear, zero biases, zero adaLN output layer (transformer.py:194-204)
    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        for blk in self.layers:
            nn.init.zeros_(blk.adaLN_modulation[-1].weight)
            nn.init.zeros_(blk.adaLN_modulation[-1].bias)

    # ---------------------------------------------------------------- HIP handle management
    def _dit_tensors(self):
        """The 49 tensors the kernels read, in t2s_dit_weights order.  Walks the modules' own parameter dicts (what
        nn.Module.__getattr__ would do 150 times per call at ~0.5 us each): the class-API loop calls this twice per
        forward, and a re-assigned Parameter is still seen because nothing is cached."""
        m = self._modules
        ts = [m["conv"]._parameters["weight"], m["conv"]._parameters["bias"], m["patch_emb"]._parameters["weight"],
              m["patch_emb"]._parameters["bias"], self._parameters["pos_embed"], m["ln"]._parameters["weight"],
              m["ln"]._parameters["bias"], m["linear_emb_to_patch"]._parameters["weight"],
              m["linear_emb_to_patch"]._parameters["bias"]]
        for blk in m["layers"]._modules.values():
            b = blk._modules
            attn, mlp, ada = b["attn"]._modules, b["mlp"]._modules, b["adaLN_modulation"]._modules["1"]._parameters
            for lin in (attn["qkv"], attn["proj"], mlp["fc1"], mlp["fc2"]):
                ts.append(lin._parameters["weight"])
                ts.append(lin._parameters["bias"])
            ts.append(ada["weight"])
            ts.append(ada["bias"])
        return ts

    def _weights_struct(self, device):
        """-> (t2s_dit_weights, the tensors it points into, stamp).  The stamp -- (data_ptr, _version) per tensor --
        changes whenever a parameter was written in place or re-allocated; while it stands, the struct of the last call
        is handed back (the reference-style loop calls the model twice per diffusion step: building 49 detached views
        and 59 ctypes fields each time cost 150 us of host time per forward, more than a small batch's kernels)."""
        ts = self._dit_tensors()
        stamp = _stamp_of(ts)
        cached = self.__dict__.get("_t2s_ws")
        if cached is not None and stamp is not None and cached[0] == device and cached[3] == stamp:
            return cached[1], cached[2], stamp
        for t in ts:
            if t.device != device:
                raise L.T2SError(f"Transformer parameters live on {t.device} but the input is on {device}; "
                                 f"call model.to(device) first")
        keep = [L.as_f32(t.detach()) for t in ts]
        for t in keep:
            L.dev_ptr(t, "Transformer parameter")          # on a GPU, fp32, contiguous -- or T2SError
        w = L.DitWeights()
        names = [n for n, _ in L.DitWeights._fields_ if n not in ("blk", "time_freqs")]
        for n, t in zip(names, keep[:9]):
            setattr(w, n, t.data_ptr())
        w.time_freqs = _freqs_on(device).data_ptr()
        bnames = [n for n, _ in L.DitBlockWeights._fields_]
        for i in range(DEPTH):
            for n, t in zip(bnames, keep[9 + 10 * i: 19 + 10 * i]):
                setattr(w.blk[i], n, t.data_ptr())
        # the C ABI carries no sizes: hand over every tensor's float count in t2s_dit_weights order (time_freqs is field 9) so an
        # undersized parameter (a foreign checkpoint assigned tensor by tensor, a sliced view) is T2S_E_INVALID with the
        # state-dict key in the message -- not an out-of-bounds read in a pack kernel
        counts = [t.numel() for t in keep[:9]] + [_freqs_on(device).numel()] + [t.numel() for t in keep[9:]]
        with torch.cuda.device(device):
            L.check(L.lib().t2s_dit_weights_check(C.byref(w), (C.c_uint64 * L.DIT_N_TENSORS)(*counts), L.DIT_N_TENSORS),
                    "t2s_dit_weights_check")
        if all(k.data_ptr() == t.data_ptr() for k, t in zip(keep, ts)):      # fp32 contiguous parameters: no copies made
            self.__dict__["_t2s_ws"] = (device, w, keep, stamp)
        return w, keep, stamp

    def t2s_handle(self, device, n_seqs: int, headroom: bool = False):
        """(Re)build or refresh the packed-weight handle: weights are re-packed whenever a parameter
        was modified in place (optimizer step, load_state_dict) or re-allocated (.to()).
        `headroom` (the training path): when the handle has to GROW, size it 12.5 % above the request -- the length groups
        of a mix-train batch fluctuate from step to step, and every rebuild is a device-wide free + multi-GB allocation."""
        device = torch.device(device)
        w, keep, stamp = self._weights_struct(device)
        h = self.__dict__.get("_t2s_h")
        if h is None or self.__dict__.get("_t2s_dev") != device or h.max_seqs < n_seqs:
            # a device synchronisation, frees and allocations: under the package's per-device lock, i.e. never while another
            # thread's sampler run has a capture open (HIP would fail this call AND invalidate that capture, _lib.device_lock)
            with L.device_lock(device):
                if h is not None:
                    h.close()
                cap = max(n_seqs, h.max_seqs if h is not None else 0)
                if headroom:
                    cap = max(cap, (n_seqs + n_seqs // 8 + 63) // 64 * 64)
                torch.cuda.synchronize(device)
                with torch.cuda.device(device):
                    h = _DitHandle(w, keep, cap)
            self.__dict__["_t2s_h"], self.__dict__["_t2s_dev"], self.__dict__["_t2s_stamp"] = h, device, stamp
            self.__dict__.pop("_t2s_math_applied", None)
        elif stamp is None or self.__dict__.get("_t2s_stamp") != stamp:
            L.check(L.lib().t2s_dit_update_weights(h.ptr, C.byref(w), L.stream_ptr(device)),
                    "t2s_dit_update_weights")
            h.keep = keep
            self.__dict__["_t2s_stamp"] = stamp
        math = self.__dict__.get("_t2s_math", "f32")
        if self.__dict__.get("_t2s_math_applied") != math:
            with L.device_lock(device), torch.cuda.device(device):       # (first bf16x3 use allocates and synchronises)
                L.check(L.lib().t2s_dit_set_math(h.ptr, L.MATH_BF16X3 if math == "bf16x3" else L.MATH_F32),
                        "t2s_dit_set_math")
            self.__dict__["_t2s_math_applied"] = math
        return h.ptr

    def t2s_handle_id(self):
        """uid of the current t2s_dit handle (None before the first use); changes whenever t2s_handle() rebuilds it."""
        h = self.__dict__.get("_t2s_h")
        return None if h is None else h.uid

    def set_math(self, math: str):
        """Matrix arithmetic of the (no-grad) forward / the sampler: "f32" (default, f32 MFMA) or "bf16x3"
        (fp32-accurate split-bf16 products on the bf16 matrix cores for the attention and the row chain,
        include/t2s.h T2S_MATH_BF16X3).  Set it before building a Sampler: a captured hipGraph keeps its kernels."""
        if math not in ("f32", "bf16x3"):
            raise ValueError(f"math must be 'f32' or 'bf16x3', got {math!r}")
        self.__dict__["_t2s_math"] = math
        return self

    def set_pairing(self, enabled: bool = True):
        """Pairing of the class-API calls `model(x_t, t, None)`, `model(x_t, t, emb)` of one diffusion step into ONE
        classifier-free-guidance pass (see _forward_nograd).  It is host-side speculation on tensor IDENTITY -- storage
        address, torch's in-place version counter, geometry -- so it cannot see a write that bypasses that counter: a
        raw-pointer kernel (this library's own in-place C entries t2s_ddpm_step / t2s_rf_step called on x_t through ctypes,
        any other ctypes / hipGraph write) or `x.data.copy_()` between the two calls.  A caller that updates x_t that way
        between the text-free and the conditional call switches pairing off here (or with T2S_NO_PAIRING=1 in the
        environment) and gets plain forwards: same results, two launch chains per step."""
        self.__dict__["_t2s_pairing"] = bool(enabled)
        self.__dict__.pop("_t2s_pair", None)
        return self

    def set_train_dtype(self, dtype: str):
        """Arithmetic of forward-under-autograd / backward: "f32" (default; gradients equal the fp32
        reference's) or "bf16" (BASELINE config 4: bf16 MFMA operands and saved activations, fp32
        accumulation, master weights, residual stream, statistics and gradients)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"train dtype must be 'f32' or 'bf16', got {dtype!r}")
        self.__dict__["_t2s_train_dtype"] = dtype
        return self

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_t2s_h", "_t2s_dev", "_t2s_stamp", "_t2s_math_applied", "_t2s_bucket", "_t2s_flat_grad", "_t2s_fwd_gen",
                  "_t2s_ws", "_t2s_pair", "_t2s_pairing"):
            state.pop(k, None)
        return state

    # ---------------------------------------------------------------- forward (transformer.py:158-193)
    def forward(self, input: torch.Tensor, t: torch.Tensor, text_input):
        """input (B,64,30) latent, t (B,) int64 (DDPM) or float (flow), text_input (B,128) or None."""
        if not input.is_cuda:
            raise L.T2SError("Transformer.forward: input must live on a GPU; the HIP path has no CPU fallback")
        if input.dim() != 3 or input.shape[1] != LAT_W or input.shape[2] != LAT_H:
            raise L.T2SError(f"Transformer.forward: input must be (B,64,30), got {tuple(input.shape)}")
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self._dit_tensors())):
            from ...train import dit_forward_autograd
            return dit_forward_autograd(self, input, t, text_input)
        return self._forward_nograd(input, t, text_input)

    def _forward_nograd(self, input, t, text_input):
        """One forward -- or, once the reference's sampling pattern has been seen, half of a PAIR.

        infer.py:79-80 / 85-86 calls `model(x_t, t, None)` and then `model(x_t, t, emb)` on the SAME x_t and t in every
        diffusion step.  Run one by one, each call is its own chain of ~10 launches (at the reference's default loader
        batch of 2 that is twice the launch floor of a step, and the unconditional branch cannot share the patchify); as
        ONE 2B-sequence classifier-free-guidance pass (t2s_dit_forward_cfg_rows) the two branches cost one chain -- and the
        results are the same bits (rows are batch-invariant; tested).  So: after a text-free call has been followed by a
        conditional call on the same (x_t, t), the next text-free call runs the pass with the remembered text and keeps
        the conditional output; the conditional call that follows hands it out if -- and only if -- x_t, t and the text are
        the very tensors (storage, version, shape) the pass read.  Anything else (another text, a changed x_t, a lone
        text-free call) falls back to plain forwards and disarms the speculation until the pattern shows again."""
        dev = input.device
        B = input.shape[0]
        x = L.as_f32(input)
        tf = L.as_f32(t.to(dev))          # `t * 100.0` promotes int64 -> fp32 in the reference too
        if tf.shape != (B,):
            raise L.T2SError(f"Transformer.forward: t must be ({B},), got {tuple(tf.shape)}")
        text = None
        if text_input is not None:
            text = L.as_f32(text_input)
            if tuple(text.shape) != (B, EMB):
                raise L.T2SError(f"Transformer.forward: text_input must be ({B},128), got {tuple(text.shape)}")
        pair = self.__dict__.setdefault("_t2s_pair", {"armed": False, "last_uncond": None, "text": None, "stash": None})
        key_x, key_t = _tensor_key(input), _tensor_key(t)
        key_text = _tensor_key(text_input) if text_input is not None else None
        # no key (a tensor made under torch.inference_mode() has no version counter), pairing switched off: never arm, never
        # hand out a stash -- plain forwards
        pairing = (key_x is not None and key_t is not None and (text_input is None or key_text is not None)
                   and self.__dict__.get("_t2s_pairing", True) and not os.environ.get("T2S_NO_PAIRING"))
        if not pairing:
            pair["armed"], pair["last_uncond"], pair["text"], pair["stash"] = False, None, None, None
        if text is not None and pair["stash"] is not None:
            # the stash HOLDS the tensors the pass read: their storage cannot have been handed to another tensor meanwhile,
            # so equal (address, version, geometry) means equal contents
            held, keys, stamp, out_c, stream_id, mode = pair["stash"]   # `held` only keeps the storages alive; `keys` were taken at the pass
            pair["stash"] = None
            del held
            if (keys == (key_x, key_t, key_text) and stamp is not None and stamp == self._param_stamp()
                    and mode == (self.t2s_handle_id(), self.__dict__.get("_t2s_math", "f32"))   # same handle, same arithmetic
                    and stream_id == torch.cuda.current_stream(dev).cuda_stream):       # same stream: ordered after the pass
                return out_c                                   # the pass of the text-free call already computed this branch
            pair["armed"] = False                              # speculation missed: back to plain forwards
        if text is None and pair["stash"] is not None:
            pair["stash"], pair["armed"] = None, False         # the last pass's conditional half was never asked for: stop guessing
        with torch.cuda.device(dev):
            speculate = (pairing and text is None and pair["armed"] and pair["text"] is not None and pair["text"].shape[0] == B
                         and pair["text"].device == dev and _tensor_key(pair["text"]) is not None)
            h = self.t2s_handle(dev, 2 * B if speculate else B)
            st = L.stream_ptr(dev)
            temb = torch.empty(B, EMB, device=dev, dtype=torch.float32)
            out = torch.empty(B, LAT_W, LAT_H, device=dev, dtype=torch.float32)
            lib = L.lib()
            L.check(lib.t2s_time_embedding(h, L.dev_ptr(tf, "t"), L.dev_ptr(temb), B, st), "t2s_time_embedding")
            if speculate:
                ptext = pair["text"]
                out_c = torch.empty_like(out)
                L.check(lib.t2s_dit_forward_cfg_rows(h, L.dev_ptr(x, "input"), L.dev_ptr(temb), B, L.dev_ptr(L.as_f32(ptext)),
                                                     L.dev_ptr(out), L.dev_ptr(out_c), B, st), "t2s_dit_forward_cfg_rows")
                pair["stash"] = ((input, t, ptext), (key_x, key_t, _tensor_key(ptext)), self.__dict__.get("_t2s_stamp"), out_c,
                                 torch.cuda.current_stream(dev).cuda_stream,
                                 (self.t2s_handle_id(), self.__dict__.get("_t2s_math", "f32")))
            else:
                L.check(lib.t2s_dit_forward(h, L.dev_ptr(x, "input"), L.dev_ptr(temb), B, L.dev_ptr(text, "text_input"),
                                            L.dev_ptr(out), B, st), "t2s_dit_forward")
        if not pairing:
            return out
        if text is None:
            pair["last_uncond"] = (key_x, key_t)
        else:
            if pair["last_uncond"] == (key_x, key_t):          # the pattern: arm, and remember THIS text tensor
                pair["armed"], pair["text"] = True, text_input
            pair["last_uncond"] = None
        return out

    def _param_stamp(self):
        return _stamp_of(self._dit_tensors())


def _stamp_of(ts):
    """(data_ptr, in-place version) per tensor, or None when torch keeps no version counter for one of them (parameters
    created under torch.inference_mode()): nothing can then be cached against the stamp -- every call re-packs."""
    if any(t.is_inference() for t in ts):
        return None
    return tuple([(t.data_ptr(), t._version) for t in ts])


def _tensor_key(t):
    """Identity of a tensor's contents as far as the host can know it: storage address, in-place version, geometry.  None
    when torch keeps no version counter for it (tensors created under torch.inference_mode(): `._version` raises) -- the
    caller then has nothing to speculate on."""
    if t.is_inference():
        return None
    return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, tuple(t.stride()))


for _cls in (Transformer, Transformerlayer, TimeEmbedding, InverseLatentEmbedding, LatentEmbedding):
    _cls.__module__ = "model.denoiser.transformer"   # pickles stay loadable by the reference and vice versa
