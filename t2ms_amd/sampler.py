"""Fused CFG sampling loop (reference infer.py:73-95) on top of t2s_sampler_*.

x_T -> steps x [2B-sequence DiT forward + CFG combine + DDPM/RF update] -> LA-VAE decode,
with one sampling step captured in a hipGraph and replayed `steps` times on a private
HIP stream.  Parity mode injects x_T and the per-step Gaussian draws; perf mode draws both
from the library's Philox stream keyed by (seed, step, GLOBAL row) so the result does not
depend on how a batch is sharded over GPUs.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .model.backbone.DDPM import ddpm_host_tables

XT_STREAM = 0xFFFFFFFF  # Philox stream id reserved for x_T

# Matrix arithmetic a Sampler selects when neither its caller nor the model's owner chose one: the fp32-ACCURATE split-bf16
# products ("bf16x3", include/t2s.h T2S_MATH_BF16X3; +35 % series/s).  Round 5 settled it with a statistics-sized table against
# an fp64 run of the oracle (profiles/r05_accuracy.md: its error is not larger than that of the reference's own PyTorch-CPU
# fp32 arithmetic at any of the 17 entries); "f32" = the exact v_mfma_f32 path, which stays the bench headline and the
# class-API default.  T2S_DEFAULT_MATH overrides (the test-suite pins f32 so the headline kernels stay covered).
DEFAULT_MATH = "bf16x3"


def default_math() -> str:
    m = os.environ.get("T2S_DEFAULT_MATH", "") or DEFAULT_MATH
    if m not in ("f32", "bf16x3"):
        raise ValueError(f"T2S_DEFAULT_MATH must be 'f32' or 'bf16x3', got {m!r}")
    return m


def loop_t_values(backbone: str, steps: int) -> torch.Tensor:
    """The t handed to the denoiser at loop index j (host, fp32).
    ddpm: floor(steps-1-j) as int64 -> fp32 (infer.py:84); flowmatching:
    round(full(j/steps)*steps)/steps in fp32 (infer.py:78)."""
    if backbone == "ddpm":
        return torch.arange(steps - 1, -1, -1, dtype=torch.int64).to(torch.float32)
    if backbone == "flowmatching":
        vals = [torch.round(torch.full((1,), j * 1.0 / steps) * steps) / steps for j in range(steps)]
        return torch.cat(vals).to(torch.float32)
    raise ValueError("No backbone found")


_STREAMS = {}


def _run_lock(device) -> "threading.RLock":
    """The package's per-device lock (_lib.device_lock): held for EVERYTHING a Sampler does on the GPU -- building its handle
    and C sampler (allocations, copies, a device synchronize), staging inputs, the run itself.  Foreign GPU work of other
    threads is not covered."""
    return L.device_lock(device)


def _sampler_stream(device) -> "torch.cuda.Stream":
    """ONE capture stream per device for every Sampler of the process.  HIP maps streams onto a few hardware queues in
    creation order and two streams on one queue run one after the other: with a stream per Sampler, lane 0 (this stream)
    landed on the queue of lane 1 (the library's pooled stream, csrc/t2s_sampler.hip lane_streams) for every fourth Sampler
    a process built -- 52 instead of 61 series/s at 64 series, reproducibly by construction order (tools/strong_probe.py)."""
    key = str(torch.device(device))
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(torch.device(device))
    return _STREAMS[key]


def _destroy_locked(device_key, ptr):
    """t2s_sampler_destroy frees device memory (a device-wide synchronisation): not while another thread's run is capturing."""
    with L.device_lock(device_key):
        L.lib().t2s_sampler_destroy(ptr)


def philox_normal(n_rows: int, row_elems: int, seed: int, stream_id: int, row0: int, device) -> torch.Tensor:
    """(n_rows, row_elems) N(0,1) draws of the library's Philox stream: element e of GLOBAL row row0 + r uses counter
    (e/4, row0 + r, stream_id), key = seed -- the same values however the rows are sharded over GPUs."""
    device = torch.device(device)
    out = torch.empty(n_rows, row_elems, device=device, dtype=torch.float32)
    if n_rows:
        with torch.cuda.device(device):
            L.check(L.lib().t2s_philox_normal(L.dev_ptr(out), int(seed), int(stream_id) & 0xFFFFFFFF, int(row0), n_rows,
                                              row_elems, L.stream_ptr(device)), "t2s_philox_normal")
    return out


def philox_uniform(n_rows: int, row_elems: int, seed: int, stream_id: int, row0: int, device) -> torch.Tensor:
    """(n_rows, row_elems) U[0,1) draws of the library's Philox stream (t2s_philox_uniform): 24-bit uniforms keyed like
    philox_normal by (seed, stream_id, GLOBAL row) -- the per-row diffusion time of a training step (train.py:109,113)."""
    device = torch.device(device)
    out = torch.empty(n_rows, row_elems, device=device, dtype=torch.float32)
    if n_rows:
        with torch.cuda.device(device):
            L.check(L.lib().t2s_philox_uniform(L.dev_ptr(out), int(seed), int(stream_id) & 0xFFFFFFFF, int(row0), n_rows,
                                               row_elems, L.stream_ptr(device)), "t2s_philox_uniform")
    return out


class Sampler:
    def __init__(self, model, decoder, backbone: str, steps: int, cfg_scale: float, batch: int, length: int,
                 device, use_graph: bool = True, seed: int = 2025, row0: int = 0, lanes: int = 0, loop_graph: int = -1,
                 math: Optional[str] = None):
        """lanes: 0 = automatic (equal part-batch chains on own streams: two when the batch is a multiple of 64 or 32 series, three for 96), 1 .. 4 -- see
        t2s_sampler_set_lanes; a scheduling choice only, the results are bitwise the same.
        math: "f32" | "bf16x3" selects the model's matrix arithmetic (Transformer.set_math) for this sampler and everything else
        that runs the model afterwards; None = what the model's owner chose with set_math, else default_math() (bf16x3)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.T2SError("Sampler needs a GPU device; the HIP path has no CPU fallback")
        self.model, self.decoder = model, decoder
        self.math = math or model.__dict__.get("_t2s_math") or default_math()
        if hasattr(model, "set_math"):
            model.set_math(self.math)
        self.backbone, self.steps, self.cfg_scale = backbone, int(steps), float(cfg_scale)
        self.batch, self.length, self.seed, self.row0 = int(batch), int(length), int(seed), int(row0)
        self.use_graph = bool(use_graph)
        self.lanes = int(lanes)
        self.loop_graph = int(loop_graph)     # 1: the whole loop as one hipGraph per lane, 0: one step replayed, -1: library default
        self.stream = _sampler_stream(self.device)
        self.ptr = None
        self._create()

    def _create(self):
        """(Re)build the C sampler against the model's CURRENT t2s_dit handle."""
        with _run_lock(self.device):
            self._create_locked()

    def _create_locked(self):
        if self.ptr is not None:
            self._fin()
        model, decoder, backbone, use_graph = self.model, self.decoder, self.backbone, self.use_graph
        tvals = loop_t_values(backbone, self.steps).contiguous()
        cfg = L.SampleConfig()
        cfg.mode = L.MODE_DDPM if backbone == "ddpm" else L.MODE_RF
        cfg.steps, cfg.cfg_scale, cfg.batch, cfg.length = self.steps, self.cfg_scale, self.batch, self.length
        cfg.use_graph, cfg.seed, cfg.row0 = int(bool(use_graph)), self.seed, self.row0
        coef = None
        if backbone == "ddpm":
            coef = ddpm_host_tables(self.steps)["coef"].contiguous()
            cfg.ddpm_coef = coef.data_ptr()          # host pointers, copied by the library
        cfg.t_values = tvals.data_ptr()
        with torch.cuda.device(self.device):
            dit = model.t2s_handle(self.device, 2 * self.batch)
            self._dit_uid = model.t2s_handle_id()
            vae = decoder._handle(self.device) if decoder is not None else None
            torch.cuda.synchronize(self.device)
            self.ptr = C.c_void_p()
            L.check(L.lib().t2s_sampler_create(dit, vae, C.byref(cfg), C.byref(self.ptr)), "t2s_sampler_create")
            L.check(L.lib().t2s_sampler_set_lanes(self.ptr, self.lanes), "t2s_sampler_set_lanes")
            L.check(L.lib().t2s_sampler_set_loop_graph(self.ptr, self.loop_graph), "t2s_sampler_set_loop_graph")
        self._fin = weakref.finalize(self, _destroy_locked, str(self.device), self.ptr)
        self._keep = (tvals, coef)

    def set_row0(self, row0: int):
        """Global index of this shard's first series for the NEXT run (the Philox key of row r is row0 + r);
        the captured hipGraph is kept (t2s_sampler_set_row0)."""
        self.row0 = int(row0)
        L.check(L.lib().t2s_sampler_set_row0(self.ptr, self.row0), "t2s_sampler_set_row0")

    def draw_xT(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_T ~ N(0,1) from the Philox stream (perf mode), (batch,64,30)."""
        if out is None:
            out = torch.empty(self.batch, L.LAT_C, L.LAT_W, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            L.check(L.lib().t2s_philox_normal(L.dev_ptr(out), self.seed, XT_STREAM, self.row0, self.batch, L.LAT,
                                              L.stream_ptr(self.device)), "t2s_philox_normal")
        return out

    def run(self, text: torch.Tensor, x_T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
            decode: bool = True, trace: bool = False):
        """Returns (latent (B,64,30), series (B,L) or None, trace (steps,L) or None).
        ``noise`` (steps,B,64,30) injects the per-step draws (parity mode)."""
        with _run_lock(self.device):
            return self._run_locked(text, x_T, noise, decode, trace)

    def _run_locked(self, text, x_T, noise, decode, trace):
        dev = self.device
        text = L.as_f32(text.to(dev))
        if tuple(text.shape) != (self.batch, L.D_MODEL):
            raise L.T2SError(f"Sampler.run: text must be ({self.batch},128), got {tuple(text.shape)}")
        # persistent device buffers: the captured hipGraph is bound to their addresses
        if self.__dict__.get("_x") is None:
            self._x = torch.empty(self.batch, L.LAT_C, L.LAT_W, device=dev, dtype=torch.float32)
            self._text = torch.empty(self.batch, L.D_MODEL, device=dev, dtype=torch.float32)
            self._series = torch.empty(self.batch, self.length, device=dev, dtype=torch.float32)
        self._text.copy_(text)
        if x_T is None:
            self.draw_xT(self._x)
        else:
            if tuple(x_T.shape) != (self.batch, L.LAT_C, L.LAT_W):
                raise L.T2SError(f"Sampler.run: x_T must be ({self.batch},64,30), got {tuple(x_T.shape)}")
            self._x.copy_(x_T)
        if noise is not None:
            noise = L.as_f32(noise.to(dev))
            if tuple(noise.shape) != (self.steps, self.batch, L.LAT_C, L.LAT_W):
                raise L.T2SError(f"Sampler.run: noise must be ({self.steps},{self.batch},64,30)")
        if (decode or trace) and self.decoder is None:
            raise L.T2SError("Sampler.run: decode requested but no decoder was given")
        tr = torch.empty(self.steps, self.length, device=dev, dtype=torch.float32) if trace else None
        with torch.cuda.device(dev):
            # weights may have changed since the last run: refresh the packed copy on the caller's stream
            self.model.t2s_handle(dev, 2 * self.batch)
            if self.model.t2s_handle_id() != self._dit_uid:
                self._create()  # the model re-created its handle (capacity grew / device moved): drop the graph
            cur = torch.cuda.current_stream(dev)
            self.stream.wait_stream(cur)
            L.check(L.lib().t2s_sampler_run(self.ptr, L.dev_ptr(self._x), L.dev_ptr(self._text), L.dev_ptr(noise),
                                            L.dev_ptr(self._series) if decode else None, L.dev_ptr(tr),
                                            self.stream.cuda_stream), "t2s_sampler_run")
            cur.wait_stream(self.stream)
        self._last = (noise, tr)  # keep caller-provided buffers alive until the stream has consumed them
        return self._x.clone(), (self._series.clone() if decode else None), tr

    @property
    def graph_lanes(self) -> int:
        """Lanes of the hipGraphs the sampler currently holds (0: nothing captured / the last run was eager)."""
        return int(L.lib().t2s_sampler_graph_lanes(self.ptr))

    def run_inplace(self, decode: bool = True):
        """Benchmark entry: x_T from Philox into the persistent buffers, no output copies.
        Requires one prior run() (buffers + text in place).  Returns (latent, series) views."""
        dev = self.device
        with _run_lock(dev), torch.cuda.device(dev):
            if self.model.t2s_handle_id() != self._dit_uid:
                self._create()
            self.draw_xT(self._x)
            cur = torch.cuda.current_stream(dev)
            self.stream.wait_stream(cur)
            L.check(L.lib().t2s_sampler_run(self.ptr, L.dev_ptr(self._x), L.dev_ptr(self._text), None,
                                            L.dev_ptr(self._series) if decode else None, None,
                                            self.stream.cuda_stream), "t2s_sampler_run")
            cur.wait_stream(self.stream)
        return self._x, self._series


def host_shard(total: int, rank: int, world: int):
    """Rows [lo,hi) of a batch of `total` series owned by `rank` (contiguous, remainder to low ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def np_save_outputs(path: str, x_1: np.ndarray, x_t: np.ndarray, lat_dec: np.ndarray, lat_enc: np.ndarray):
    """The four arrays evaluation.py reads (infer.py:118-123)."""
    import os
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "x_1.npy"), x_1[:, :, np.newaxis])
    np.save(os.path.join(path, "x_t.npy"), x_t[:, :, np.newaxis])
    np.save(os.path.join(path, "x_t_latent_dec_array.npy"), lat_dec)
    np.save(os.path.join(path, "x_t_latent_enc_array.npy"), lat_enc)
