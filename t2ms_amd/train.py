"""Training-side host glue of the T2S path (reference train.py:101-136): autograd bridge to the HIP
training kernels, MSE loss, fused AdamW with torch-compatible state, flat-bucket gradient all-reduce.

All arithmetic is in libt2s_hip.so (t2s_dit_train_forward/_backward, t2s_mse[_backward],
t2s_adamw_step); torch provides the autograd graph, the optimizer/scheduler bookkeeping and
torch.distributed (RCCL) for the one gradient all-reduce per step.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

N_GRAD = 925592   # values that receive a gradient: DiT minus pos_embed minus unpatch.*


# ---------------------------------------------------------------------------- loss
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
    # the reduction's partials live in the CALLER's scratch (stream-ordered torch memory): no state in the library, so
    # losses computed on other streams / by other threads cannot meet
    scratch = torch.empty(L.MSE_SCRATCH_FLOATS, device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):
        L.check(L.lib().t2s_mse_ws(L.dev_ptr(ac), L.dev_ptr(bc), out.data_ptr(), ac.numel(), L.dev_ptr(scratch),
                                   L.stream_ptr(a.device)), "t2s_mse_ws")
    return out


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _mse_value(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ac, bc, gc = L.as_f32(a.detach()), L.as_f32(b.detach()), L.as_f32(g.detach()).reshape(1)
        da = torch.empty_like(ac) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(bc) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(a.device):
            L.check(L.lib().t2s_mse_backward(L.dev_ptr(ac), L.dev_ptr(bc), L.dev_ptr(gc), L.dev_ptr(da),
                                             L.dev_ptr(db), ac.numel(), L.stream_ptr(a.device)), "t2s_mse_backward")
        return da, db


# ---------------------------------------------------------------------------- DiT forward/backward
_GRAD_TOP = ("conv_w", "conv_b", "patch_w", "patch_b", None, "ln_w", "ln_b", "out_w", "out_b")   # None = pos_embed
_GRAD_BLK = ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ada_w", "ada_b")


BUCKET = N_GRAD + 8   # the flat gradient bucket; slot N_GRAD carries the (weighted) loss through the same all-reduce


def _trainable(model):
    """The 48 tensors that receive a gradient, in bucket order (= _dit_tensors() minus pos_embed)."""
    ts = model._dit_tensors()
    return [t for i, t in enumerate(ts) if i != 4]


def _make_bucket(model, dev):
    """A flat fp32 bucket + one view per gradient tensor + the t2s_dit_grads struct pointing into it."""
    flat = torch.zeros(BUCKET, device=dev, dtype=torch.float32)
    g = L.DitGrads()
    views, off = [], 0
    ts = _trainable(model)
    for name, t in zip([n for n in _GRAD_TOP if n is not None], ts[:8]):
        n = t.numel()
        v = flat[off:off + n].view(t.shape)
        setattr(g, name, v.data_ptr())
        views.append(v)
        off += n
    for i in range(4):
        for j, name in enumerate(_GRAD_BLK):
            t = ts[8 + 10 * i + j]
            n = t.numel()
            v = flat[off:off + n].view(t.shape)
            setattr(g.blk[i], name, v.data_ptr())
            views.append(v)
            off += n
    assert off == N_GRAD, off
    return dict(flat=flat, views=views, struct=g)


def grad_bucket(model, dev):
    """The model's PERSISTENT gradient bucket on `dev`: every backward writes here, p.grad aliases its views, the
    DDP all-reduce sends it as one message, and the fused AdamW's pointer table stays valid from step to step."""
    b = model.__dict__.get("_t2s_bucket")
    if b is None or b["flat"].device != dev:
        b = _make_bucket(model, dev)
        model.__dict__["_t2s_bucket"] = b
    return b


class _DitTrainFn(torch.autograd.Function):
    """pred = Transformer(input, t, text) with a hand-written backward (train.py:123-125)."""

    @staticmethod
    def forward(ctx, model, x, temb, text, *params):
        dev = x.device
        B = x.shape[0]
        with torch.cuda.device(dev):
            h = model.t2s_handle(dev, B, headroom=True)     # refreshes the packed weights if they changed
            w, keep, _ = model._weights_struct(dev)
            dt = L.TRAIN_BF16 if model.__dict__.get("_t2s_train_dtype", "f32") == "bf16" else L.TRAIN_F32
            L.check(L.lib().t2s_dit_set_train_dtype(h, dt), "t2s_dit_set_train_dtype")
            out = torch.empty(B, L.LAT_C, L.LAT_W, device=dev, dtype=torch.float32)
            L.check(L.lib().t2s_dit_train_forward(h, C.byref(w), L.dev_ptr(x, "input"), L.dev_ptr(temb), B,
                                                  L.dev_ptr(text, "text_input"), L.dev_ptr(out), B,
                                                  L.stream_ptr(dev)), "t2s_dit_train_forward")
        # the saved activations live in the model's ONE handle: a later grad-mode forward overwrites them
        gen = model.__dict__.get("_t2s_fwd_gen", 0) + 1
        model.__dict__["_t2s_fwd_gen"] = gen
        ctx.model, ctx.B, ctx.keep, ctx.gen, ctx.handle_id = model, B, keep, gen, model.t2s_handle_id()
        return out

    @staticmethod
    def backward(ctx, dout):
        model, B = ctx.model, ctx.B
        dev = dout.device
        if model.__dict__.get("_t2s_fwd_gen") != ctx.gen or model.t2s_handle_id() != ctx.handle_id:
            raise L.T2SError("Transformer backward: another grad-mode forward ran on this model (or its handle was "
                             "rebuilt) since the forward being differentiated -- the saved activations are gone. "
                             "Run forward -> backward pairs one at a time (wrap extra forwards in torch.no_grad()).")
        bucket = grad_bucket(model, dev)
        ts = _trainable(model)
        if any(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(ts[:1], bucket["views"][:1])):
            # gradients of an earlier backward still live in the persistent bucket (accumulation, or
            # zero_grad(set_to_none=False)): write this pass elsewhere so autograd can add the two
            bucket = _make_bucket(model, dev)
        model.__dict__["_t2s_flat_grad"] = bucket["flat"]
        d = L.as_f32(dout)
        with torch.cuda.device(dev):
            h = model.t2s_handle(dev, B)
            L.check(L.lib().t2s_dit_train_backward(h, L.dev_ptr(d, "grad_output"), C.byref(bucket["struct"]), B,
                                                   L.stream_ptr(dev)), "t2s_dit_train_backward")
        # fresh view objects: AccumulateGrad adopts a returned gradient in place of copying it only when nothing
        # else references the tensor object -- p.grad must ALIAS the bucket (all-reduce, AdamW pointer table)
        views = [v.view_as(v) for v in bucket["views"]]
        grads = views[:4] + [None] + views[4:]      # pos_embed (index 4 of _dit_tensors) gets none
        dx = None
        if ctx.needs_input_grad[1]:                 # something upstream of the latent trains (un-frozen LA-VAE encoder)
            dx = torch.empty(B, L.LAT_C, L.LAT_W, device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                L.check(L.lib().t2s_dit_train_input_grad(h, L.dev_ptr(dx), B, L.stream_ptr(dev)), "t2s_dit_train_input_grad")
        return (None, dx, None, None) + tuple(grads)


def dit_forward_autograd(model, input, t, text_input):
    """Transformer.forward under autograd (called from the mirror's forward when grads are on)."""
    dev = input.device
    B = input.shape[0]
    x = L.as_f32(input)            # a latent that requires grad (un-frozen encoder) gets its gradient from t2s_dit_train_input_grad
    tf = L.as_f32(t.to(dev))
    if tf.shape != (B,):
        raise L.T2SError(f"Transformer.forward: t must be ({B},), got {tuple(tf.shape)}")
    text = None
    if text_input is not None:
        text = L.as_f32(text_input.detach())
        if tuple(text.shape) != (B, L.D_MODEL):
            raise L.T2SError(f"Transformer.forward: text_input must be ({B},128), got {tuple(text.shape)}")
    temb = model.time_emb(tf)
    return _DitTrainFn.apply(model, x, temb, text, *model._dit_tensors())


# ---------------------------------------------------------------------------- optimizer
class T2SAdamW(torch.optim.Optimizer):
    """AdamW whose update runs in one fused HIP kernel per tensor (t2s_adamw_step).  Defaults and the
    per-parameter state keys (step, exp_avg, exp_avg_sq) equal torch.optim.AdamW's, so
    optimizer.state_dict() / load_state_dict() interchange with the reference's checkpoints
    (train.py:37,44,94)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise L.T2SError("T2SAdamW: parameters must live on a GPU")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                todo.append((p, L.as_f32(p.grad), st))
            if not todo:
                continue
            steps = {int(st["step"].item()) for _, _, st in todo}
            dev = todo[0][0].device
            same_dev = all(p.device == dev for p, _, _ in todo)
            if len(steps) == 1 and same_dev and len(todo) <= 64:
                # one launch for the whole group: device table of (param, grad, exp_avg, exp_avg_sq, n), rebuilt
                # only when a pointer changed (the flat gradient bucket is re-allocated by every backward)
                rows = [(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                        for p, g, st in todo]
                key = tuple(rows)
                cache = group.setdefault("_t2s_table", {})
                if cache.get("key") != key:
                    cache["key"] = key
                    cache["dev"] = torch.tensor(rows, dtype=torch.int64).to(dev)      # 5 x 8 bytes = t2s_adamw_tensor
                    cache["chunks"] = sum((r[4] + 1023) // 1024 for r in rows)
                with torch.cuda.device(dev):
                    L.check(lib.t2s_adamw_step_multi(cache["dev"].data_ptr(), len(rows), cache["chunks"], float(group["lr"]),
                                                     float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                     steps.pop(), L.stream_ptr(dev)), "t2s_adamw_step_multi")
            else:
                for p, g, st in todo:
                    with torch.cuda.device(p.device):
                        L.check(lib.t2s_adamw_step(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                   st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]),
                                                   float(b1), float(b2), float(group["eps"]),
                                                   float(group["weight_decay"]), int(st["step"].item()),
                                                   L.stream_ptr(p.device)), "t2s_adamw_step")
            # the kernels wrote the parameters in place behind autograd's back: bump the version counters so the
            # mirror's packed-weight cache (keyed on data_ptr + _version) notices
            for p, _, _ in todo:
                torch.autograd.graph.increment_version(p)
        return loss

    def load_state_dict(self, state_dict):
        """torch.optim.Optimizer.load_state_dict leaves `step` on whatever device torch.load(map_location=...) put it
        (it is moved only for fused / capturable optimizers): a resumed run would then bump 48 one-element GPU tensors
        and read them back with .item() -- 48 host syncs -- on every step.  `step` is host bookkeeping here, as in a
        fresh run (torch.tensor(0.0)): keep it on the CPU."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].device.type != "cpu":
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_t2s_table", None)      # device pointers: never part of a checkpoint
        return sd


# ---------------------------------------------------------------------------- data parallel
def allreduce_param_grads(params, dist, n_local: Optional[int] = None, n_global: Optional[int] = None,
                          loss: Optional[torch.Tensor] = None):
    """The same collective for a plain torch module (the MLP denoiser of BASELINE configs[0]: gradients from t2s_mlp_backward or torch autograd, no persistent
    bucket): the gradients of `params` are flattened into ONE message, weighted n_local / n_global, summed over the ranks
    and scattered back; a parameter without a gradient (empty shard) contributes zeros.  Returns the global mean loss."""
    if dist is None:
        return loss
    params = [p for p in params if p.requires_grad]
    dev = params[0].device
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params] +
                     [(loss.detach() if loss is not None else torch.zeros((), device=dev)).reshape(1).float()])
    world = dist.get_world_size()
    flat.mul_((float(n_local) / float(n_global)) if (n_local is not None and n_global) else 1.0 / world)
    from . import dist as tdist
    tdist.all_reduce_sum(dist, flat)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p).to(p.dtype)
        off += n
    return flat[off].clone()


def allreduce_gradients(model, dist, n_local: Optional[int] = None, n_global: Optional[int] = None,
                        loss: Optional[torch.Tensor] = None):
    """Combine the DiT gradients of all ranks with ONE all-reduce of the flat 3.7 MB bucket (SURVEY.md 8e:
    latency-bound, a single bucket beats stock DDP's many).  Rank r's gradient of ITS mean loss over n_local rows is
    weighted n_local / n_global, so the sum is the gradient of the mean over the global batch even when shards are
    ragged (default weights: 1 / world).  A rank whose shard is empty (p.grad is None after zero_grad) contributes a
    zero bucket -- every rank must call this every step.  `loss` rides in the bucket's spare slot.  Returns
    (flat bucket, global mean loss or None).  No-op when `dist` is None."""
    ts = _trainable(model)
    if dist is None:
        return model.__dict__.get("_t2s_flat_grad"), loss
    dev = ts[0].device
    bucket = grad_bucket(model, dev)
    flat, views = bucket["flat"], bucket["views"]
    if all(p.grad is None for p in ts):
        flat.zero_()
    for p, v in zip(ts, views):
        if p.grad is None:
            p.grad = v
        elif p.grad.data_ptr() != v.data_ptr():       # autograd cloned / accumulated elsewhere: move it into the bucket
            v.copy_(p.grad)
            p.grad = v
    world = dist.get_world_size()
    w = (float(n_local) / float(n_global)) if (n_local is not None and n_global) else 1.0 / world
    flat[N_GRAD] = loss.detach() if loss is not None else 0.0
    flat.mul_(w)
    from . import dist as tdist
    tdist.all_reduce_sum(dist, flat)
    model.__dict__["_t2s_flat_grad"] = flat
    return flat, (flat[N_GRAD].clone() if loss is not None or n_local == 0 else None)
