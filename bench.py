#!/usr/bin/env python3
"""Headline benchmark: generated series / second for 1000-step CFG DDPM sampling with the DiT
denoiser (BASELINE.json configs[1]: B=256 per GPU, L=96, cfg 9.0, fp32), synthetic inputs.

One "step" = one pass of the hot path over one batch: x_T (Philox, on device) -> 1000 x
[512-sequence DiT forward + CFG combine + DDPM update] -> LA-VAE decode to (256,96).  Inputs
(text embeddings, weights) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: `value` is WEAK scaling -- every rank samples its own 256-series shard (global rows [256*rank, 256*rank+256) of
the Philox stream); no data-path collective, RCCL only for the barrier and the max-over-ranks time.  BASELINE's metric read
literally is STRONG scaling (256 series at 1/2/4/8 GPUs), so the line also carries
  * at N = 1: `strong_shards` -- the per-GPU share of that job (128 / 64 / 32 series) timed on this GPU, with the predicted
    strong efficiency rate(256/N) / rate(256) (no collective: N x rate(256/N) is the N-GPU rate);
  * at N > 1: `strong` -- the same 256 series split over the ranks (their union equals the one-GPU batch bit for bit).

The same JSON line carries, as extra keys: `roofline` (dominant kernel, in-situ HIP-event timing), `roofline_rows` (the three
row-chain kernel instances, same timing), `cpu_baseline` (the oracle on the host cores, N=1 only: 3 warm + 10 measured CFG steps),
`config3` / `config5` (BASELINE configs[2] and configs[4] under this clock, N=1 only), `alt_math` (the bf16x3 arithmetic with its
own roofline against the bf16 peak / 6 and an in-run `accuracy_vs_fp64` block, N=1 only),
`train` (BASELINE configs[3] shape: the bf16 DiT training step of train.py at B=1152 per GPU with the gradient all-reduce at
N>1; its `roofline` prints mfma_frac, the designed and the measured HBM fraction and bytes_vs_minimal; `mix_train_shard` = the
384-row step of an 8-GPU mix-train run) and the DRIVER legs (N=1; at N>1 on --legs): `infer_driver` = infer.py itself at the
authors' flags, `train_driver` = train.py's own mix-train loop, `class_api` = the reference-style loop against the mirrors.

    python bench.py --gpus N        # without a torchrun environment: starts the N ranks itself (spawn_ranks)
"""
import argparse
import contextlib
import json
import os
import sys
import time
import types

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# algorithmic FLOPs
FLOP_ATTN_PER_SEQ_BLOCK = 2 * 2 * 4 * 480 * 480 * 32          # QK^T + PV, 4 heads = 117.96 MFLOP
FLOP_FORWARD_PER_SEQ = 0.977e9
PEAK_FP32_MFMA_TFLOPS = 157.3                                 # MI355X_MICROARCH.md, Peak FP32 (matrix)


def build_models(dev, seed=2025):
    from t2ms_amd import synth
    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    m = Transformer()
    m.load_state_dict(synth.make_dit_state_dict(seed), strict=True)   # adaLN re-initialised N(0,0.02)
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                    embedding_dim=64))
    v.load_state_dict(synth.make_vae_state_dict(seed), strict=True)
    m.set_math("f32")        # explicit: every figure of this file is on the exact f32 MFMA unless it says bf16x3 (alt_math, infer_driver)
    return m.to(dev).eval(), v.to(dev).eval()


def time_kernels_in_situ(model, dev, x, text, n_steps=8):
    """Average launch duration of the dominant kernel (fused attention) measured IN SITU: HIP
    events recorded by the library on the launching stream around every kernel of real CFG
    forwards (same shapes, same data path as the timed region, eager launches)."""
    import ctypes as C
    from t2ms_amd import _lib as L
    B = x.shape[0]
    lib = L.lib()
    with torch.cuda.device(dev):
        h = model.t2s_handle(dev, 2 * B)
        st = torch.cuda.current_stream(dev).cuda_stream
        temb = model.time_emb(torch.full((1,), 500, device=dev))
        ou, oc = torch.empty_like(x), torch.empty_like(x)
        for _ in range(2):
            L.check(lib.t2s_dit_forward_cfg(h, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(),
                                            oc.data_ptr(), B, st))
        torch.cuda.synchronize(dev)
        L.check(lib.t2s_dit_timing_begin(h))
        for _ in range(n_steps):
            L.check(lib.t2s_dit_forward_cfg(h, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(),
                                            oc.data_ptr(), B, st))
        out = (C.c_double * 22)()
        L.check(lib.t2s_dit_timing_end_ex(h, out, 11))       # classes 9 / 10: the first / last row-chain launch (also in class 1)
    mid_ms, mid_n = out[2] - out[18] - out[20], out[3] - out[19] - out[21]
    return {"attn_us": out[0] / out[1] * 1e3, "attn_calls": int(out[1]),
            "rows_us": out[2] / out[3] * 1e3, "rows_calls": int(out[3]),
            "rows_first_us": out[18] / max(out[19], 1) * 1e3, "rows_mid_us": mid_ms / max(mid_n, 1) * 1e3,
            "rows_last_us": out[20] / max(out[21], 1) * 1e3, "rows_mid_calls": int(mid_n),
            "other_us": out[4] / max(out[5], 1) * 1e3, "other_calls": int(out[5]),
            "forward_us": (out[0] + out[2] + out[4]) / n_steps * 1e3}


# algorithmic FLOPs of the three row-chain instances per sequence (480 tokens x 128 inputs x output columns x 2)
FLOP_ROWS_PER_SEQ = {
    "dit_rows_kernel<false,true> (block 0: patchify prologue + LN1.mod + qkv)": 2 * 480 * 128 * 384 + 2 * 480 * (16 + 512),
    "dit_rows_kernel<true,true> (x3 per forward: proj + MLP of block i, LN1.mod + qkv of block i+1)": 2 * 480 * 128 * (128 + 256 + 256 + 384),
    "dit_rows_kernel<true,false> (block 3: proj + MLP + final LayerNorm / Linear 128->4 / unpatchify)": 2 * 480 * 128 * (128 + 256 + 256 + 4),
}


def roofline_rows(kt, n_seq):
    """The row chain is 53 % of a sampling step: its three kernel instances against the fp32 MFMA peak, from the same in-situ
    HIP-event timing as `roofline` (each kernel alone on the chip, 512-sequence CFG passes)."""
    us = dict(zip(FLOP_ROWS_PER_SEQ, (kt["rows_first_us"], kt["rows_mid_us"], kt["rows_last_us"])))
    inst, flops, t = {}, 0.0, 0.0
    for (name, per_seq), mult in zip(FLOP_ROWS_PER_SEQ.items(), (1, 3, 1)):
        f = per_seq * n_seq
        ach = f / (us[name] * 1e-6) / 1e12
        inst[name] = {"avg_launch_us": us[name], "flop_per_launch": f, "achieved": ach, "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                      "launches_per_forward": mult}
        flops += mult * f
        t += mult * us[name] * 1e-6
    return {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "instances": inst,
            "achieved": flops / t / 1e12, "frac": flops / t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "timing": "HIP events around every row-chain launch of the same 8 eager 512-sequence CFG forwards as `roofline`"}


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup cpu quota, 16).
    os.cpu_count() reports the whole host (256 on the GPU box) although a one-GPU job owns a
    16-core share; oversubscribing torch's pool there is 17x slower than using the share."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("T2S_CPU_BASELINE_THREADS", "16"))))


def _cpu_cfg_steps(O, sd, x, text, tab, diff_steps, cfg, n_steps, g, n_warm=1):
    """Wall time per CFG step (2 DiT forwards + DDPM update) of the oracle, after n_warm warm steps."""
    batch = x.shape[0]

    def one_step(x, j):
        t = torch.full((batch,), diff_steps - 1 - j, dtype=torch.long)
        u = O.dit_forward(sd, x, t, None)
        c = O.dit_forward(sd, x, t, text)
        return O.ddpm_p_sample(tab, x, u + cfg * (c - u), t, torch.randn(x.shape, generator=g))

    with torch.no_grad():
        for j in range(n_warm):
            x = one_step(x, j)
        t0 = time.perf_counter()
        for j in range(n_warm, n_warm + n_steps):
            x = one_step(x, j)
        return (time.perf_counter() - t0) / n_steps, x


def cpu_baseline(batch, diff_steps, cfg, length, n_cfg_steps=10, n_warm=3, one_thread_batch=16):
    """The CPU oracle (torch fp32) on a bounded sample of the same workload, as SURVEY.md section 8(d) words it: n_warm
    warm + n_cfg_steps measured CFG steps at the full batch on this job's host cores + one decode, extrapolated to
    diff_steps steps (every step costs the same); plus the
    single-thread figure BASELINE.md section 3 asks for, on a smaller batch (series/s on a CPU is flat in the batch:
    SURVEY.md section 6 measured 0.110 at B=32 and 0.114 at B=256).  Attention runs as F.scaled_dot_product_attention --
    what timm 1.0.11's Attention.forward (fused_attn) executes in the reference -- not the oracle's explicit softmax."""
    from oracle import t2s_oracle as O
    from t2ms_amd import synth
    cores = usable_cores()
    sd = synth.make_dit_state_dict(2025)
    vsd = synth.make_vae_state_dict(2025)
    text = synth.make_text_embeddings(2025, batch)
    tab = O.ddpm_tables(diff_steps)
    g = torch.Generator().manual_seed(0)
    O.set_attention_impl("sdpa")
    try:
        torch.set_num_threads(cores)
        t_step, x = _cpu_cfg_steps(O, sd, synth.make_latents(2025, batch), text, tab, diff_steps, cfg, n_cfg_steps, g,
                                   n_warm=n_warm)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.vae_decode(vsd, x, length)
            t_dec = time.perf_counter() - t0
        torch.set_num_threads(1)
        b1 = min(batch, one_thread_batch)
        t1_step, x1 = _cpu_cfg_steps(O, sd, synth.make_latents(2025, b1), text[:b1], tab, diff_steps, cfg, 1, g)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.vae_decode(vsd, x1, length)
            t1_dec = time.perf_counter() - t0
    finally:
        O.set_attention_impl("explicit")
        torch.set_num_threads(cores)
    total = t_step * diff_steps + t_dec
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": batch / total, "unit": "series/s", "cores": cores, "kind": "port",
            "sample": f"{n_warm} warm + {n_cfg_steps} measured CFG steps (2 DiT forwards + DDPM update) at B={batch} + 1 decode on {cores} threads, "
                      f"extrapolated x{diff_steps}/{n_cfg_steps}; {t_step:.3f} s/step; attention = "
                      f"F.scaled_dot_product_attention (the reference's timm fused_attn path); CPU: {cpu_model}",
            "one_thread": {"value": b1 / (t1_step * diff_steps + t1_dec), "unit": "series/s", "cores": 1,
                           "sample": f"1 CFG step at B={b1} + 1 decode on 1 thread, extrapolated x{diff_steps}; "
                                     f"{t1_step:.3f} s/step"}}


# ---------------------------------------------------------------------------- training leg (BASELINE configs[3])
FLOP_TRAIN_PER_SAMPLE = 3 * 0.977e9            # forward + backward = 3 x the forward
TRAIN_CLASSES = ("attention", "row_chain", "other", "train_gemm", "train_attn_fwd", "train_attn_bwd", "train_wgrad",
                 "train_elementwise", "train_tail")


def train_leg(dev, dist, rank, world, batch=1152, length=96, steps=30, warmup=4, dtype="bf16", mix_shard=True):
    """BASELINE configs[3] shape under the same clock discipline as the headline: DiT training step of train.py
    (train.train_step: cached LA-VAE latents -> q_sample -> forward -> MSE -> backward -> ONE flat-bucket all-reduce
    -> fused AdamW), bf16 MFMA operands with fp32 master weights, per-GPU batch 1152 (global 9216 on 8 GPUs), L=96,
    DDPM T=100, synthetic rows.  Every rank runs it; the time is the max over ranks."""
    import ctypes as C
    import train as drv
    from t2ms_amd import _lib as L
    from t2ms_amd import dist as tdist
    from t2ms_amd import latent_cache, synth
    from t2ms_amd.train import T2SAdamW
    from model.backbone.DDPM import DDPM
    model, vae = build_models(dev)
    model.train().set_train_dtype(dtype)
    model.encoder = vae.encoder
    for n, p in model.named_parameters():
        if "encoder" in n:
            p.requires_grad = False
    opt = T2SAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    args = types.SimpleNamespace(backbone="ddpm", total_step=100, seed=2025)
    n_global = batch * world
    x = synth.make_series(1, batch, length)                       # this rank's rows; the driver slices a global batch
    text = synth.make_text_embeddings(1, batch)
    lat_local = latent_cache.encode_all(model.encoder, x, dev)
    # train_step shards a GLOBAL batch by rank: hand it global-shaped tables whose rows [lo, hi) are this rank's data --
    # the resident form train.py itself uses (device tables + a device index vector: no host -> device copy in a step)
    lo, _ = tdist.shard_rows(n_global, rank, world)
    eg = torch.zeros(n_global, 128, device=dev)
    eg[lo:lo + batch] = text.to(dev)
    latents = torch.zeros(n_global, 64, 30, device=dev)
    latents[lo:lo + batch] = lat_local
    idx = torch.arange(n_global, device=dev)
    torch.manual_seed(2025)

    def run(n, d, first):
        loss = None
        for i in range(n):
            loss = drv.train_step(model, ddpm, opt, d, args, None, None, dev, rank, world, latents, idx, first + i, eg)
        return loss

    run(warmup, dist, 0)
    tdist.barrier(dist, dev)
    t0 = time.perf_counter()
    loss = run(steps, dist, warmup)
    tdist.barrier(dist, dev)
    el = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    out = {"metric": "DiT training samples/sec (configs[3] shape: bf16, B=1152/GPU, L=96, DDPM T=100, cached latents)",
           "value": n_global * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "warmup": warmup, "per_gpu_batch": batch, "global_batch": n_global, "dtype": dtype,
           "tflops_algorithmic": FLOP_TRAIN_PER_SAMPLE * n_global * steps / el / 1e12,
           "loss": float(loss.detach()), "optimizer": "fused AdamW lr 1e-4 (t2s_adamw_step_multi)", "data": "synthetic"}
    if dist is not None:        # the same steps with the collective skipped: its share of the step
        tdist.barrier(dist, dev)
        t0 = time.perf_counter()
        run(steps, None, warmup + steps)
        tdist.barrier(dist, dev)
        el0 = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
        out["ms_per_step_without_allreduce"] = el0 / steps * 1e3
        out["allreduce_share"] = max(0.0, 1.0 - el0 / el)
        out["allreduce"] = "one SUM all-reduce of the flat 3.7 MB fp32 gradient bucket per step (RCCL)"
    if rank == 0 and dist is None and batch == 1152 and mix_shard:
        # the shape the reference's DEFAULT flags train at (mix-train, train.py:145): a 9,216-row batch is three length
        # groups of ~3,072 rows = 384 rows per GPU and step on eight GPUs -- the step at that shard size, same model
        t0 = time.perf_counter()
        sub = torch.arange(384, device=dev)
        for i in range(4):
            drv.train_step(model, ddpm, opt, None, args, None, None, dev, 0, 1, latents, sub, 20_000 + i, eg)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(20):
            drv.train_step(model, ddpm, opt, None, args, None, None, dev, 0, 1, latents, sub, 20_010 + i, eg)
        torch.cuda.synchronize(dev)
        ms384 = (time.perf_counter() - t0) / 20 * 1e3
        out["mix_train_shard"] = {"rows_per_gpu_and_step": 384, "ms_per_step": ms384, "samples_per_s_per_gpu": 384 / ms384 * 1e3,
                                  "frac_of_1152_row_rate": (384 / ms384) / (batch / (el / steps * 1e3)),
                                  "note": "one GPU on the per-GPU shard of an 8-GPU mix-train step (three length groups per "
                                          "9,216-row batch); no all-reduce in this figure"}
    if rank == 0:               # per-class kernel time, in situ (HIP events around every launch of 3 eager steps)
        h = model.t2s_handle(dev, batch)
        torch.cuda.synchronize(dev)
        L.check(L.lib().t2s_dit_timing_begin(h))
        n_t = 3
        run(n_t, None, 10_000)
        buf = (C.c_double * 18)()
        L.check(L.lib().t2s_dit_timing_end_ex(h, buf, 9))
        br = {TRAIN_CLASSES[i]: {"ms_per_step": buf[2 * i] / n_t, "launches_per_step": buf[2 * i + 1] / n_t}
              for i in range(3, 9) if buf[2 * i + 1] > 0}
        out["kernel_classes"] = br
        top = max(br, key=lambda k: br[k]["ms_per_step"])
        bytes_model = train_bytes_model(batch)
        out["hbm_bytes_per_step_model"] = bytes_model["total"]
        out["hbm_bytes_model_note"] = bytes_model["note"]
        kernel_ms = sum(v["ms_per_step"] for v in br.values())
        traffic = _train_traffic_from_profile(batch)
        t_k = kernel_ms * 1e-3
        tflops = FLOP_TRAIN_PER_SAMPLE * batch / t_k / 1e12
        minimal = TRAIN_MINIMAL_U * bytes_model["u"]
        # three readings of the same step, none hidden: SURVEY 8(d) names the bf16 MFMA peak as config 4's roofline
        # (mfma_frac); the kernels are built HBM-bound, so `frac` prices the bytes they move BY DESIGN against 8 TB/s and
        # hbm_frac_measured prices the bytes the PMC passes actually counted; bytes_vs_minimal says how far the design is
        # from the byte-minimal one of DESIGN.md 4.3 (one recompute chain per half block, weight gradients inside the chains)
        out["roofline"] = {"bound": "hbm", "kernel": f"whole step (largest class: {top})",
                           "achieved": bytes_model["total"] / t_k / 1e9, "peak": 8000.0, "unit": "GB/s",
                           "frac": bytes_model["total"] / t_k / 1e9 / 8000.0,
                           "traffic": traffic,
                           "hbm_frac_measured": (traffic / t_k / 1e9 / 8000.0) if traffic else None,
                           "bytes_minimal": minimal,
                           "bytes_vs_minimal": (traffic / minimal) if traffic else bytes_model["total"] / minimal,
                           "mfma_tflops": tflops, "mfma_peak_tflops": PEAK_BF16_MFMA_TFLOPS,
                           "mfma_frac": tflops / PEAK_BF16_MFMA_TFLOPS,
                           "traffic_source": {"file": "profiles/train_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                      "passes of tools/bench_train.py)", "measured_by_this_run": False},
                           "traffic_note": "FETCH_SIZE counts Infinity-Cache hits (MI355X_MICROARCH.md, HBM): the forward's re-reads "
                                           "of what the previous launch just wrote are such hits -- removing all 21 u of them changes the "
                                           "step by +0.16 % (profiles/r05_chain_bound_ab.txt), so bytes_vs_minimal overstates what "
                                           "byte-cutting can buy; DESIGN.md 4.3 'structural floor'",
                           "timing": f"sum of HIP-event launch durations over {n_t} eager steps = {kernel_ms:.2f} ms/step"}
    return out


# Bytes the bf16 training step moves BY DESIGN, per launch of each kernel, in units u = one bf16 (M,128) tensor
# (M = batch * 480 tokens; an fp32 (M,128) tensor is 2 u).  Kept next to the launch list of t2s_dit_train_forward /
# _backward (csrc/t2s_train.hip): update both together.  DESIGN.md section 4.3 has the same table.
TRAIN_U = {
    # forward, per block
    "qkv GEMM, LN prologue fused with the previous block's gate/residual add (x 2 + branch 1 in; x 2 + a1 1 + q,k,v 3 out)": (9, 4),
    "  ... block 0 has no add to fuse (x 2 in; a1 1 + q,k,v 3 out)": (-3, 1),
    "attention forward (q,k,v in; o out)": (4, 4),
    "proj GEMM": (2, 4),
    "fc1 GEMM, LN prologue fused with the attention gate/residual add (x 2 + p 1 in; x_mid 2 + a2 1 + u 2 out)": (8, 4),
    "fc2 GEMM, GELU prologue (u 2 in; f 1 out)": (3, 4),
    # backward, per block
    "fc2 weight gradient (df 1, u 2)": (3, 4), "fc2 data gradient x gelu' (df 1, u 2 in; du 2 out)": (5, 4),
    "fc1 weight gradient (du 2, a2 1)": (3, 4),
    "fc1 data gradient with LN2 backward + attention-branch gate backward as its epilogue (du 2, x 2, dx 2, p 1 in; dx 2, dp 1 out)": (10, 4),
    "proj weight gradient (dp 1, o 1)": (2, 4), "proj data gradient": (2, 4),
    "attention D_i (o 1, do 1)": (2, 4), "attention dQ (q,k,v,do in; dq out)": (5, 4), "attention dK,dV (q,k,v,do in; dk,dv out)": (6, 4),
    "qkv weight gradient (dqkv 3, a1 1)": (4, 4),
    "qkv data gradient with LN1 backward + next gate backward as its epilogue (dqkv 3, x 2, dx 2, f 1 in; dx 2, df 1 out)": (11, 4),
    "  ... block 0 has no gate to differentiate next (no f in, no df out)": (-2, 1),
    "column-sum partials of the two fused LayerNorm backwards (M/32 x 384 fp32 written + re-read)": (0.375, 8),
    "weight-gradient partial tiles written + re-read (4 gradients)": (2.6, 4),
    # tails
    "patchify (x 2 out) + final layer forward (x_mid 2 + f 1 in: the last gate/residual add is formed in the kernel)": (5, 1),
    "final layer backward with the last MLP gate backward (x_mid 2 + f 1 in; dx 2 + df 1 out) + patchify backward (dx 2 in)": (8, 1),
}


TRAIN_MINIMAL_U = 168       # 42 u per block for the byte-minimal chain design = 23.8 GB at B = 1152
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak


def train_bytes_model(batch):
    u = batch * 480 * 128 * 2
    units = sum(per * n for per, n in TRAIN_U.values())
    return {"total": units * u, "u": u, "note": f"{units:.1f} u per step by design (table TRAIN_U in bench.py / DESIGN.md 4.3), "
                                        f"u = {u / 1e6:.1f} MB = one bf16 (M,128) tensor"}


def _train_traffic_from_profile(batch):
    """HBM bytes per step from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (None if absent)."""
    f = os.path.join(REPO, "profiles", "train_traffic.json")
    try:
        d = json.load(open(f))
        return d.get("hbm_bytes_per_step") if d.get("per_gpu_batch") == batch else None
    except (OSError, ValueError):
        return None


def strong_shards(model, vae, args, dev, rate_full):
    """BASELINE's metric read literally is B = 256 series at 1/2/4/8 GPUs ("256 total for strong
    scaling -- report both").  Sampling has no collective, so the strong curve is decided by how fast ONE GPU samples its
    shard of 256/N series: the same workload at B = 128, 64, 32 (one timed batch each after a warm / capture batch).
    predicted_strong_efficiency[N] = rate(256/N) / rate(256) (series/s per GPU relative to the full batch);
    predicted_strong_speedup[N] = N x that."""
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    rates = {args.batch: rate_full}
    out = {"shards": {}}
    for world in (2, 4, 8):
        n = args.batch // world
        s = Sampler(model, vae.decoder, args.backbone, args.diffusion_steps, args.cfg_scale, n, args.length, dev,
                    use_graph=not args.no_graph, seed=2025, row0=0, lanes=args.lanes)
        s.run(synth.make_text_embeddings(2025, n).to(dev), decode=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        s.run_inplace(decode=True)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        rates[n] = n / el
        out["shards"][str(n)] = {"series_per_s": n / el, "ms_per_batch": el * 1e3, "gpus_for_256_total": world}
        del s
    out["predicted_strong_efficiency"] = {str(w): rates[args.batch // w] / rate_full for w in (2, 4, 8)}
    out["predicted_strong_speedup"] = {str(w): w * rates[args.batch // w] / rate_full for w in (2, 4, 8)}
    out["predicted_strong_series_per_s"] = {str(w): w * rates[args.batch // w] for w in (2, 4, 8)}
    out["note"] = ("one GPU, the per-GPU shard of a 256-series job at N GPUs; no data-path collective, so N x rate(256/N) is "
                   "the N-GPU strong-scaling rate up to launch skew between ranks")
    return out


def strong_leg(model, vae, args, dev, dist, rank, world):
    """N > 1: the SAME 256 series as the one-GPU headline, split over the ranks (rows [lo, hi) of the global Philox
    stream, so the union equals the N = 1 batch bit for bit) -- strong scaling next to the weak `value`."""
    from t2ms_amd import dist as tdist
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    lo, hi = tdist.shard_rows(args.batch, rank, world)
    n = hi - lo
    s = Sampler(model, vae.decoder, args.backbone, args.diffusion_steps, args.cfg_scale, max(n, 1), args.length, dev,
                use_graph=not args.no_graph, seed=2025, row0=lo, lanes=args.lanes)
    text = synth.make_text_embeddings(2025, args.batch)[lo:hi].contiguous().to(dev)
    if n:
        s.run(text, decode=True)
    tdist.barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if n:
            s.run_inplace(decode=True)
    tdist.barrier(dist, dev)
    el = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    return {"value": args.batch * args.steps / el, "unit": "series/s", "scaling": "strong", "global_batch": args.batch,
            "per_gpu_batch": args.batch // world, "ms_per_step": el / args.steps * 1e3, "steps": args.steps}


def infer_driver_leg(dev, dist, rank, world, rows=2048):
    """infer.py ITSELF at the flags the authors script (reference scripts/script.sh:4: `infer.py --dataset_name
    exchangerate_24 --cfg_scale 7.0 --total_step 100`, i.e. flowmatching, the default --batch_size 2) on `rows` synthetic
    test rows with seeded weights: dataset + loader order + encode + coalesced sampler launches + decode + the final
    gather + D2H, and separately the whole main() incl. model construction, graph capture and writing the four .npy
    files.  Beside it the same shape as ONE resident sampler (B = 256, 100 RF steps, cfg 7): what the kernels allow."""
    import shutil
    import tempfile
    import infer as drv
    from t2ms_amd import dist as tdist
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    tmp = tempfile.mkdtemp(prefix="t2s_bench_infer_") if rank == 0 else None
    tmp = tmp if dist is None else _bcast_obj(dist, tmp)
    argv = ["--dataset_name", "exchangerate_24", "--cfg_scale", "7.0", "--total_step", "100", "--synthetic", str(rows),
            "--random_init", "--seed", "2025", "--save_path", tmp, "--no_figs"]
    try:
        tdist.barrier(dist, dev)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):        # stdout carries the ONE result line only
            a = drv.main(argv)
        tdist.barrier(dist, dev)
        wall = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
        st = a.stats
    finally:
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    out = {"metric": "infer.py at the authors' flags (scripts/script.sh:4): flowmatching, 100 steps, cfg 7, --batch_size 2, L=24",
           "math": a.math, "value": st["series"] / st["loop_s"], "unit": "series/s", "series": st["series"], "loop_s": st["loop_s"],
           "whole_main_s": wall, "whole_main_series_per_s": st["series"] / wall, "launches": st["launches"],
           "series_per_launch_and_gpu": st["series_per_launch_and_gpu"], "loader_batch": st["loader_batch"], "n_gpus": world,
           "data": f"synthetic ({rows} test rows)",
           "note": "loop = resident test split -> encode -> coalesced sampler launches -> decode -> final gather -> host; "
                   "whole_main adds model construction, hipGraph capture and np.save of the four files (the ten jpg plots skipped)"}
    if rank == 0:       # the kernels' own rate at this shape: one resident 256-series sampler, same steps / cfg / length
        model, vae = build_models(dev)
        s = Sampler(model, vae.decoder, "flowmatching", 100, 7.0, 256, 24, dev, use_graph=True, seed=2025, row0=0, math=a.math)
        s.run(synth.make_text_embeddings(2025, 256).to(dev), decode=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            s.run_inplace(decode=True)
        torch.cuda.synchronize(dev)
        ref = 3 * 256 / (time.perf_counter() - t0)
        out["resident_sampler_b256_series_per_s"] = ref
        out["frac_of_resident_sampler"] = out["value"] / (ref * world)
    return out


def class_api_leg(dev, batch=256, length=96, total_steps=1000, cfg=9.0, warm=10, timed=100):
    """The loop of reference infer.py:76-88 written against the mirrored classes -- per diffusion step two
    `model(input=, t=, text_input=)` calls, the CFG combination as torch glue, `ddpm.p_sample` -- i.e. the literal drop-in
    of INTEGRATION.md section 1, eager launches, no fused sampler (the mirror pairs the two calls into one CFG pass).  `timed` steps after `warm`, extrapolated to
    `total_steps` (every step costs the same) + one decode; beside it the fused sampler as ONE lane (same shape)."""
    from model.backbone.DDPM import DDPM
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    model, vae = build_models(dev)
    ddpm = DDPM(total_steps, dev)
    emb = synth.make_text_embeddings(2025, batch).to(dev)
    x_t = torch.randn(batch, 64, 30, device=dev)

    def steps(x_t, j0, n):
        for j in range(j0, j0 + n):
            t = torch.full((x_t.size(0),), total_steps - 1 - j, dtype=torch.long, device=dev)
            pred_uncond = model(input=x_t, t=t, text_input=None)
            pred_cond = model(input=x_t, t=t, text_input=emb)
            pred = pred_uncond + cfg * (pred_cond - pred_uncond)
            x_t = ddpm.p_sample(x_t, pred, t)
        return x_t

    with torch.no_grad():
        x_t = steps(x_t, 0, warm)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        x_t = steps(x_t, warm, timed)
        t_host = time.perf_counter() - t0            # the host has ENQUEUED the steps: its own cost per step
        torch.cuda.synchronize(dev)
        t_step = (time.perf_counter() - t0) / timed
        t0 = time.perf_counter()
        series, _ = vae.decoder(x_t, length=length)
        torch.cuda.synchronize(dev)
        t_dec = time.perf_counter() - t0
    assert bool(torch.isfinite(series).all())
    one = Sampler(model, vae.decoder, "ddpm", 100, cfg, batch, length, dev, use_graph=True, seed=2025, row0=0, lanes=1)
    one.run(emb, decode=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(2):
        one.run_inplace(decode=False)
    torch.cuda.synchronize(dev)
    t_fused = (time.perf_counter() - t0) / 200
    value = batch / (t_step * total_steps + t_dec)
    return {"metric": f"reference-style class-API loop (infer.py:76-88 against the mirrors), B={batch}, {total_steps}-step DDPM, cfg {cfg}",
            "value": value, "unit": "series/s", "ms_per_diffusion_step": t_step * 1e3,
            "host_enqueue_ms_per_diffusion_step": t_host / timed * 1e3,
            "fused_sampler_one_lane_ms_per_diffusion_step": t_fused * 1e3, "frac_of_fused_one_lane": t_fused / t_step,
            "sample": f"{warm} warm + {timed} timed steps, extrapolated x{total_steps}/{timed} + 1 decode ({t_dec * 1e3:.1f} ms)"}


def _bcast_obj(dist, obj, src=0):
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def train_driver_leg(dev, dist, rank, world, rows_per_length=200_000, batch=9216, warm_batches=2, timed_batches=8):
    """train.py's OWN loop (reference train.py:52-95 shape: mix-train, the default --batch_size 9216 split by the collate
    into three length groups per batch) on 3 x `rows_per_length` synthetic rows, latent cache on, bf16, DDPM T=100: loader
    order, row gathers, the rank's slice, q_sample, forward, loss, backward, all-reduce, AdamW, loss bookkeeping -- the
    clock starts after `warm_batches` loader batches and stops after `timed_batches` more (device drained at both ends)."""
    import shutil
    import tempfile
    import train as drv
    from t2ms_amd import dist as tdist
    tmp = tempfile.mkdtemp(prefix="t2s_bench_train_")
    argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--batch_size", str(batch), "--epochs", "1", "--save_path", tmp,
            "--synthetic", str(rows_per_length), "--random_init", "--checkpoint_path", "", "--seed", "2025", "--bf16",
            "--max_steps", str(3 * (warm_batches + timed_batches))]
    a = drv.get_args(argv)
    a.max_steps += 3                 # three more steps after the clock stops, event-timed kernel by kernel
    clock = {}

    def on_step(n, n_rows, model):
        import ctypes as C
        from t2ms_amd import _lib as L
        if n == 3 * warm_batches:
            torch.cuda.synchronize(dev)
            clock["t0"], clock["rows"] = time.perf_counter(), 0
        elif 3 * warm_batches < n <= 3 * (warm_batches + timed_batches):
            clock["rows"] += n_rows
            if n == 3 * (warm_batches + timed_batches):
                torch.cuda.synchronize(dev)
                clock["t1"] = time.perf_counter()
                L.check(L.lib().t2s_dit_timing_begin(model.t2s_handle(dev, 1)))      # is a step the sum of its kernels?
        elif n == 3 * (warm_batches + timed_batches) + 3:
            buf = (C.c_double * 18)()
            L.check(L.lib().t2s_dit_timing_end_ex(model.t2s_handle(dev, 1), buf, 9))
            clock["kernel_ms"] = sum(buf[2 * i] for i in range(3, 9)) / 3

    a.on_step = on_step
    try:
        with contextlib.redirect_stdout(sys.stderr):
            losses = drv.train(a)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    el = tdist.max_over_ranks(dist, clock["t1"] - clock["t0"], dev)
    steps = 3 * timed_batches
    return {"metric": "train.py's own loop: mix-train, --batch_size 9216 (three length groups per batch), latent cache, bf16, DDPM T=100",
            "value": clock["rows"] / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3, "steps": steps,
            "rows_per_step": clock["rows"] / steps, "rows_per_step_and_gpu": clock["rows"] / steps / world,
            "kernel_ms_per_step": clock.get("kernel_ms"),
            "kernels_note": "sum of HIP-event launch durations of the DiT training kernels over the 3 steps after the clock stopped: "
                            "ms_per_step well above it means the loop stalled on the host side, both high means slow kernels",
            "dataset_rows": 3 * rows_per_length, "n_gpus": world, "loss": losses[-1] if losses else None,
            "data": "synthetic", "data_path": "resident tables (datafactory.epoch_index_batches; train.py --loader_batches walks the DataLoader instead)"}


def alt_math_run(model, vae, args, dev, text):
    """One extra batch of the same workload in bf16x3 arithmetic (include/t2s.h T2S_MATH_BF16X3: fp32-accurate,
    six bf16 MFMAs per product, attention and row chain).  Reported NEXT TO the headline, never as it."""
    from t2ms_amd.sampler import Sampler
    model.set_math("bf16x3")
    try:
        s2 = Sampler(model, vae.decoder, args.backbone, args.diffusion_steps, args.cfg_scale, args.batch, args.length,
                     dev, use_graph=not args.no_graph, seed=2025, row0=0, lanes=args.lanes)
        s2.run(text, decode=True)                 # captures its own graph with the x3 kernels
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        s2.run_inplace(decode=True)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        kt = time_kernels_in_situ(model, dev, torch.randn(args.batch, 64, 30, device=dev), text)
        shards = None
        if not args.no_strong and args.batch % 8 == 0 and args.batch >= 64:
            # the drivers' default arithmetic on the per-GPU shards of a 256-series job (as `strong_shards` does for f32)
            shards = strong_shards(model, vae, args, dev, args.batch / el)
    finally:
        model.set_math("f32")
    # its own roofline: an fp32-accurate product costs SIX bf16 MFMAs, so the ceiling of this arithmetic is the dense bf16
    # matrix peak / 6 in algorithmic (fp32-equivalent) FLOPs; `achieved` counts the algorithmic FLOPs of the attention kernel
    ach = FLOP_ATTN_PER_SEQ_BLOCK * 2 * args.batch / (kt["attn_us"] * 1e-6) / 1e12
    whole = FLOP_FORWARD_PER_SEQ * 2 * args.batch * args.diffusion_steps / el / 1e12
    return {"math": "bf16x3: every product of the attention and the row chain as six bf16 MFMAs, fp32-accurate (include/t2s.h T2S_MATH_BF16X3)",
            "value": args.batch / el, "unit": "series/s", "ms_per_step": el * 1e3,
            "attention_us": kt["attn_us"], "row_chain_us": kt["rows_us"],
            "attention_bf16_tflops_executed": 6 * ach, "bf16_dense_peak_tflops": PEAK_BF16_MFMA_TFLOPS,
            "strong_shards": shards,
            "roofline": {"bound": "mfma", "kernel": "attn_x3_kernel (six bf16 MFMAs per fp32-accurate product)", "achieved": ach,
                         "peak": PEAK_BF16_MFMA_TFLOPS / 6, "unit": "TFLOP/s (algorithmic, fp32-equivalent)",
                         "frac": ach / (PEAK_BF16_MFMA_TFLOPS / 6), "avg_launch_us": kt["attn_us"],
                         "whole_path_tflops": whole, "whole_path_frac": whole / (PEAK_BF16_MFMA_TFLOPS / 6)}}


def config3_leg(dev, batch=1024, steps=100, cfg=5.0, length=96, timed=2, math="f32"):
    """BASELINE configs[2] under this clock: rectified flow, B = 1024, 100 steps (infer.py:135 default), cfg 5 (the traffic
    rows of scripts/script.sh:31-33), the WHOLE loop of each lane in one hipGraph: one warm (capturing) batch + `timed` batches."""
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    model, vae = build_models(dev)            # its own handle (2048 sequences): the headline's samplers keep theirs
    s = Sampler(model, vae.decoder, "flowmatching", steps, cfg, batch, length, dev, use_graph=True, seed=2025, row0=0, loop_graph=1,
                math=math)
    s.run(synth.make_text_embeddings(2025, batch).to(dev), decode=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(timed):
        lat, series = s.run_inplace(decode=True)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    assert bool(torch.isfinite(series).all())
    tflops = FLOP_FORWARD_PER_SEQ * 2 * batch * steps * timed / el / 1e12
    return {"metric": f"generated series/sec (configs[2]: B={batch}, L={length}, {steps}-step rectified flow, cfg {cfg}, whole loop in one hipGraph per lane)",
            "value": batch * timed / el, "unit": "series/s", "ms_per_batch": el / timed * 1e3, "batches_timed": timed,
            "graph_lanes": s.graph_lanes, "whole_path_tflops": tflops, "whole_path_frac_of_fp32_mfma_peak": tflops / PEAK_FP32_MFMA_TFLOPS,
            "dtype": "f32", "math": math, "data": "synthetic"}


def config5_leg(dev, model, vae, sampler96, text, batch=256, steps=1000, cfg=9.0):
    """BASELINE configs[4] under this clock: variable-length groups L in {24, 48, 96} (the collate's length groups,
    dataloader.py:115-133), each END TO END on one GPU -- LA-VAE encode of the (B, L) series (infer.py:73-74), the 1000-step
    CFG DDPM loop on the (B,64,30) latent, LA-VAE decode to (B, L): one timed batch per length after that length's
    capturing batch (L = 96 reuses the headline's warm sampler).  The DiT works on the (64,30) latent whatever L is, so the
    three rates differ only by the codec."""
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    out, total_t = {"lengths": {}}, 0.0
    for L_ in (24, 48, 96):
        if L_ == 96:
            s = sampler96
        else:
            s = Sampler(model, vae.decoder, "ddpm", steps, cfg, batch, L_, dev, use_graph=True, seed=2025, row0=0)
            s.run(text, decode=True)
        x1 = synth.make_series(7, batch, L_).to(dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.no_grad():
            z_enc, _ = vae.encoder(x1)                      # infer.py:73-74 (model.encoder IS pretrained_model.encoder, :47)
        lat, series = s.run_inplace(decode=True)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        assert tuple(z_enc.shape) == (batch, 64, 30) and tuple(series.shape) == (batch, L_) and bool(torch.isfinite(series).all())
        out["lengths"][str(L_)] = {"series_per_s": batch / el, "ms_per_batch": el * 1e3}
        total_t += el
        if L_ != 96:
            del s
    tflops = FLOP_FORWARD_PER_SEQ * 2 * batch * steps * 3 / total_t / 1e12
    out.update({"metric": f"generated series/sec (configs[4]: encode + {steps}-step DDPM + decode, B={batch} per length group, L in 24/48/96)",
                "value": 3 * batch / total_t, "unit": "series/s", "whole_path_tflops": tflops,
                "whole_path_frac_of_fp32_mfma_peak": tflops / PEAK_FP32_MFMA_TFLOPS, "dtype": "f32", "data": "synthetic",
                "sample": "one timed batch per length, each after a capturing batch of the same sampler"})
    return out


def accuracy_vs_fp64(dev, batch=32):
    """The evidence beside `alt_math`: ONE conditional forward at B = 32, t = 500, in the three fp32
    arithmetics -- f32 MFMA, bf16x3, and the CPU fp32 oracle (the reference's PyTorch-CPU arithmetic) -- against the fp64
    arithmetic of the oracle on the same fp32 data (tools/accuracy_table.py is the full table: 768 rows per entry, chains,
    the 1000-step taps; profiles/r05_accuracy.md).  Part of the cpu_baseline leg: the oracle is the checker here too."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import numpy as np
    from accuracy_table import dbl, fp64_arithmetic
    from oracle import t2s_oracle as O
    from t2ms_amd import synth
    from model.denoiser.transformer import Transformer
    sd = synth.make_dit_state_dict(2025)
    x, text = synth.make_latents(41, batch), synth.make_text_embeddings(41, batch)
    t = torch.full((batch,), 500, dtype=torch.long)
    with torch.no_grad():
        cpu32 = O.dit_forward(sd, x, t, text)
        with fp64_arithmetic():
            ref = O.dit_forward(dbl(sd), x.double(), t, text.double())
        cols = {"cpu_fp32_oracle": cpu32.double()}
        for math, key in (("f32", "f32_mfma"), ("bf16x3", "bf16x3")):
            m = Transformer()
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).eval().set_math(math)
            cols[key] = m(input=x.to(dev), t=t.to(dev), text_input=text.to(dev)).cpu().double()
            del m
    res = {}
    for k, v in cols.items():
        d = (v - ref).abs()
        res[k] = {"rms": float(np.sqrt(float((d ** 2).mean()))), "max_abs": float(d.max())}
    res["bf16x3_over_oracle"] = {"rms": res["bf16x3"]["rms"] / res["cpu_fp32_oracle"]["rms"],
                                 "max_abs": res["bf16x3"]["max_abs"] / res["cpu_fp32_oracle"]["max_abs"]}
    res["f32_mfma_over_oracle"] = {"rms": res["f32_mfma"]["rms"] / res["cpu_fp32_oracle"]["rms"],
                                   "max_abs": res["f32_mfma"]["max_abs"] / res["cpu_fp32_oracle"]["max_abs"]}
    res["sample"] = (f"one conditional forward, B={batch} (61,440 outputs), t=500; reference = fp64 arithmetic of the oracle on the "
                     f"fp32 weights / inputs / time embedding; max |ref| {float(ref.abs().max()):.2f}")
    return res


def pmc_probe(dev):
    """`bench.py --pmc-probe` (the CHILD of measure_pmc, under `rocprofv3 --pmc ...`): a few eager 512-sequence CFG forwards at
    the headline shape, every kernel alone on the chip (the launch shape of `roofline`), nothing else."""
    from t2ms_amd import _lib as L
    from t2ms_amd import synth
    model, _ = build_models(dev)
    B = 256
    x = synth.make_latents(3, B).to(dev)
    text = synth.make_text_embeddings(2025, B).to(dev)
    lib = L.lib()
    with torch.cuda.device(dev):
        h = model.t2s_handle(dev, 2 * B)
        st = torch.cuda.current_stream(dev).cuda_stream
        temb = model.time_emb(torch.full((1,), 500, device=dev))
        ou, oc = torch.empty_like(x), torch.empty_like(x)
        for _ in range(4):
            L.check(lib.t2s_dit_forward_cfg(h, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(), oc.data_ptr(), B, st))
        torch.cuda.synchronize(dev)


def measure_pmc(passes, timeout_s=90):
    """HBM bytes / matrix-busy cycles of the hot kernels measured BY THIS RUN: one `rocprofv3 --pmc <counters>` child per pass
    (separate passes for FETCH_SIZE and WRITE_SIZE, as MI355X_MICROARCH.md prescribes; counters only, no trace domains) around
    `bench.py --pmc-probe`.  The parent has initialised the GPU, so the profiler is STARTED AS A CHILD (never an exec), with the
    interpreter itself after `--`.  -> {kernel name: {counter: average per dispatch}}; {} if rocprofv3 is missing or a pass fails
    (the caller then falls back to the committed passes and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    out = {}
    for counters in passes:
        tmp = tempfile.mkdtemp(prefix="t2s_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--pmc-probe"]
            # its own process group: a pass that overruns is ended WITH the profiled grandchild (an exact group we started)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                print(f"bench.py: rocprofv3 --pmc {' '.join(counters)} overran {timeout_s} s and was ended", file=sys.stderr)
                return {}
            files = sorted(glob.glob(os.path.join(tmp, "*", "*counter_collection.csv")))
            if proc.returncode != 0 or not files:
                print(f"bench.py: rocprofv3 --pmc {' '.join(counters)} failed (rc {proc.returncode}): {(err or '')[-300:]}", file=sys.stderr)
                return {}
            agg = {}
            for row in csv.DictReader(open(files[-1])):
                a = agg.setdefault((row["Kernel_Name"], row["Counter_Name"]), [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
            for (k, c), (v, n) in agg.items():
                out.setdefault(k, {})[c] = v / n
        except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
            print(f"bench.py: PMC pass {counters} failed: {type(e).__name__}: {e}", file=sys.stderr)
            return {}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def _pmc_of(pm, part, counter):
    for k, d in pm.items():
        if part in k and counter in d:
            return d[counter]
    return None


def spawn_ranks(n_gpus, argv=None):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, one FRESH process per
    GPU (`python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a CHILD -- never an exec: this
    parent has made no GPU call, and it stays a plain launcher that relays rank 0's JSON line and the child's exit
    code).  Rendezvous on 127.0.0.1 at a port the kernel just handed out."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)    # stderr passes through
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")               # rank 0's ONE result line, printed once the job has ended cleanly
        else:
            sys.stdout.write(ln)
            sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed batches (each = full 1000-step sampling of B series)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="series per GPU")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--cfg-scale", type=float, default=9.0)
    ap.add_argument("--length", type=int, default=96)
    ap.add_argument("--backbone", default="ddpm", choices=["ddpm", "flowmatching"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="sampler lanes: 0 = the library's default (two half-batch chains on two streams when B is a multiple of 64 or B = 32; three at 96), 1 .. 4")
    ap.add_argument("--loop-graph", type=int, default=-1, choices=[-1, 0, 1],
                    help="what a lane's hipGraph holds: 1 = the whole loop (steps x 10 kernel nodes, one launch per run), 0 = one "
                         "step replayed `steps` times, -1 = the library's default (include/t2s.h t2s_sampler_set_loop_graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg reported as `train`")
    ap.add_argument("--no-strong", action="store_true",
                    help="skip the strong-scaling figures (N = 1: `strong_shards`, the per-GPU shards 128 / 64 / 32 of a "
                         "256-series job timed on this GPU; N > 1: `strong`, the 256 series split over the ranks)")
    ap.add_argument("--legs", action="store_true",
                    help="run the driver legs at N > 1 too (default: N = 1 only -- a leg is a whole driver run with collectives of "
                         "its own, and the scaling record should not depend on it)")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the driver legs: `infer_driver` (infer.py at the authors' flags), `train_driver` (train.py's own "
                         "mix-train loop) and `class_api` (the reference-style loop against the mirrored classes)")
    ap.add_argument("--infer-driver-rows", type=int, default=2048, help="synthetic test rows of the `infer_driver` leg")
    ap.add_argument("--train-driver-rows", type=int, default=200_000, help="synthetic rows PER LENGTH of the `train_driver` leg")
    ap.add_argument("--train-driver-batch", type=int, default=9216, help="--batch_size of the `train_driver` leg (train.py:142)")
    ap.add_argument("--train-batch", type=int, default=1152, help="per-GPU batch of the training leg")
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--no-alt-math", action="store_true", help="skip the extra bf16x3 measurement reported as alt_math")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip `config3` (BASELINE configs[2]: rectified flow, B=1024, 100 steps, whole-loop graph) and `config5` "
                         "(configs[4]: encode + 1000-step DDPM + decode at L = 24 / 48 / 96)")
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3"],
                    help="matrix arithmetic: f32 MFMA (headline) or fp32-accurate split-bf16 products (include/t2s.h)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the in-run rocprofv3 --pmc passes (roofline.traffic / mfma_busy then come from the committed profiles)")
    ap.add_argument("--pmc-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_probe:
        torch.cuda.set_device(0)
        pmc_probe(torch.device("cuda", 0))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))       # plain `python bench.py --gpus N`: this process becomes the launcher
    if world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    from t2ms_amd import dist as tdist
    from t2ms_amd import synth
    local_rank = tdist.local_device_index()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from t2ms_amd.sampler import Sampler
    dist = tdist.init("nccl", dev)      # RCCL; None when single-process

    B = args.batch
    model, vae = build_models(dev)
    model.set_math(args.math)
    sampler = Sampler(model, vae.decoder, args.backbone, args.diffusion_steps, args.cfg_scale, B, args.length,
                      dev, use_graph=not args.no_graph, seed=2025, row0=rank * B, lanes=args.lanes, loop_graph=args.loop_graph)
    text = synth.make_text_embeddings(2025, B, row0=rank * B).to(dev)
    sampler.run(text, decode=True)                      # allocates persistent buffers, captures the graph
    for _ in range(max(0, args.warmup - 1)):
        sampler.run_inplace(decode=True)

    tdist.barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lat, series = sampler.run_inplace(decode=True)
    tdist.barrier(dist, dev)
    elapsed = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    assert bool(torch.isfinite(series).all()) and bool(torch.isfinite(lat).all())

    total_series = args.steps * B * world
    lanes_used = sampler.graph_lanes or args.lanes or 1       # what the library captured for this batch size
    value = total_series / elapsed
    out = {
        "metric": "generated series/sec (B=256, L=96, 1000-step DDPM)",
        "value": value, "unit": "series/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{'configs[1]' if (args.backbone, args.diffusion_steps, B) == ('ddpm', 1000, 256) else 'custom'}: DiT denoiser, {args.diffusion_steps}-step {args.backbone} with CFG "
                               f"(2 forwards/step), B={B}/GPU, L={args.length}, cfg_scale={args.cfg_scale}, "
                               f"LA-VAE decode; Philox noise on device; hipGraph={'off' if args.no_graph else 'on'}; "
                               f"sampler lanes={lanes_used} (part-batch chains on own streams, include/t2s.h t2s_sampler_set_lanes)",
                   "global_batch": B * world, "diffusion_steps": args.diffusion_steps, "parallelism": f"batch-shard x{world}",
                   "sampler_lanes": lanes_used},
    }
    strong = None
    if dist is not None and not args.no_strong:
        strong = strong_leg(model, vae, args, dev, dist, rank, world)
    train = None
    if not args.no_train:
        try:
            train = train_leg(dev, dist, rank, world, batch=args.train_batch, steps=args.train_steps)
        except Exception as e:          # the headline line must still be printed; the failure is reported in it
            import traceback
            traceback.print_exc()
            train = {"error": f"{type(e).__name__}: {e}"}
    legs = {}
    if not args.no_legs and (world == 1 or args.legs):
        for name, fn in (("infer_driver", lambda *a: infer_driver_leg(*a, rows=args.infer_driver_rows)),
                         ("train_driver", lambda *a: train_driver_leg(*a, rows_per_length=args.train_driver_rows,
                                                                     batch=args.train_driver_batch))):
            try:
                legs[name] = fn(dev, dist, rank, world)
            except Exception as e:      # the headline line must still be printed; the failure is reported in it
                import traceback
                traceback.print_exc()
                legs[name] = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0 and world == 1 and (args.backbone, B) == ("ddpm", 256):
            try:
                legs["class_api"] = class_api_leg(dev, B, args.length, args.diffusion_steps, args.cfg_scale)
            except Exception as e:
                import traceback
                traceback.print_exc()
                legs["class_api"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        # roofline of the dominant kernel (fused attention; 48 % of all FLOPs), same shapes as the workload
        kt = time_kernels_in_situ(model, dev, lat.clone(), text)
        t_attn = kt["attn_us"] * 1e-6
        flop_attn = FLOP_ATTN_PER_SEQ_BLOCK * 2 * B
        achieved = flop_attn / t_attn / 1e12
        # HBM traffic / matrix-busy share of that kernel: PMC counters need rocprofv3 passes of their own -- taken by THIS run as
        # child processes around the same launch shape (measure_pmc; FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE x 2
        # on gfx950, KB units: MI355X_MICROARCH.md "HBM"); if the profiler is not available, the committed passes, labelled
        pm = {}
        if world == 1 and B == 256 and not args.no_pmc:
            pm = measure_pmc([["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]])
        fa, wa = _pmc_of(pm, "attn_fwd_persistent", "FETCH_SIZE"), _pmc_of(pm, "attn_fwd_persistent", "WRITE_SIZE")
        busy = _pmc_of(pm, "attn_fwd_persistent", "SQ_VALU_MFMA_BUSY_CYCLES")
        if fa is not None and wa is not None:
            traffic = (2 * fa + wa) * 1024
            pmc = {"mfma_busy_frac": (busy / 1024 / (t_attn * 2.4e9)) if busy else None, "measured_by_this_run": True,
                   "fetch_size_kb_raw": fa, "write_size_kb": wa, "mfma_busy_cycles_per_dispatch": busy,
                   "source": "rocprofv3 --pmc passes started by this run around `bench.py --pmc-probe` (4 eager 512-sequence CFG forwards); "
                             "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / "
                             "(in-situ launch duration x 2.4 GHz)"}
        else:
            pmc = {}
            pfile = os.path.join(REPO, "profiles", "attn_pmc.json")
            if os.path.exists(pfile) and B == 256:     # the committed PMC passes were taken at the headline shape
                try:
                    pmc = json.load(open(pfile))
                    pmc["measured_by_this_run"] = False
                except (OSError, ValueError):
                    pmc = {}
            traffic = pmc.get("hbm_bytes_per_launch")
        out["roofline"] = {"bound": "mfma", "kernel": "attn_fwd_persistent_kernel", "achieved": achieved,
                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                           "traffic": traffic, "hbm_gbps": (traffic / t_attn / 1e9) if traffic else None,
                           "pmc": {k: pmc.get(k) for k in ("mfma_busy_frac", "measured_by_this_run", "fetch_size_kb_raw", "write_size_kb",
                                                           "source")} if pmc else None,
                           "avg_launch_us": t_attn * 1e6,
                           "flop_per_launch": flop_attn, "launches_timed": kt["attn_calls"],
                           "timing": "HIP events on the launch stream around every attention launch of "
                                     "8 eager 512-sequence CFG forwards (in situ), the kernel alone on the chip -- the "
                                     "launch shape of `--lanes 1`, whose rocprofv3 average agrees "
                                     "(profiles/*_kernel_stats_lanes1.csv). With 2 lanes the timed region issues this kernel "
                                     "as two 256-sequence launches that time-share the CUs with the other lane's kernels; "
                                     "per-launch durations there are not a kernel property, the pipelined figure is "
                                     "whole_path_frac_of_fp32_mfma_peak"}
        out["roofline_rows"] = roofline_rows(kt, 2 * B)
        if pm:   # the row-chain instances' HBM bytes and matrix-busy cycles from the same passes
            rows_pm = {}
            for k, d in pm.items():
                if "dit_rows_kernel" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                    kk = k.replace(" ", "")
                    inst = next((i for i in ("<false,true>", "<true,true>", "<true,false>") if i in kk or
                                 {"<false,true>": "ILb0ELb1E", "<true,true>": "ILb1ELb1E", "<true,false>": "ILb1ELb0E"}[i] in kk), kk[:40])
                    rows_pm[inst] = {"traffic": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
                                     "mfma_busy_cycles_per_dispatch": d.get("SQ_VALU_MFMA_BUSY_CYCLES")}
            out["roofline_rows"]["pmc"] = {"instances": rows_pm, "measured_by_this_run": True}
        out["kernel_breakdown_us"] = {"attention_x4": kt["attn_us"], "row_chain_x5": kt["rows_us"],
                                      "other_x1_adaln": kt["other_us"], "forward_total": kt["forward_us"]}
        # whole-step figure for context: all DiT FLOPs / wall time
        step_flops = FLOP_FORWARD_PER_SEQ * 2 * B * args.diffusion_steps * args.steps
        out["whole_path_tflops"] = step_flops / elapsed / 1e12
        out["whole_path_frac_of_fp32_mfma_peak"] = out["whole_path_tflops"] / PEAK_FP32_MFMA_TFLOPS
        if strong is not None:
            out["strong"] = strong
        if world == 1 and not args.no_strong and B % 8 == 0 and B >= 64:
            out["strong_shards"] = strong_shards(model, vae, args, dev, value)
        if world == 1 and not args.no_configs and (args.backbone, args.diffusion_steps, B, args.length) == ("ddpm", 1000, 256, 96):
            for name, fn in (("config5", lambda: config5_leg(dev, model, vae, sampler, text, B, args.diffusion_steps, args.cfg_scale)),
                             ("config3", lambda: config3_leg(dev))):
                try:
                    out[name] = fn()
                except Exception as e:      # the headline line must still be printed; the failure is reported in it
                    import traceback
                    traceback.print_exc()
                    out[name] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.math == "f32" and not args.no_alt_math:
            out["alt_math"] = alt_math_run(model, vae, args, dev, text)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, args.diffusion_steps, args.cfg_scale, args.length)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            if "alt_math" in out:
                if not args.no_configs:
                    try:      # BASELINE configs[2] in the drivers' default arithmetic
                        c3 = config3_leg(dev, math="bf16x3")
                        out["alt_math"]["config3"] = {k: c3[k] for k in ("metric", "value", "unit", "ms_per_batch", "math")}
                    except Exception as e:
                        out["alt_math"]["config3"] = {"error": f"{type(e).__name__}: {e}"}
                try:
                    out["alt_math"]["accuracy_vs_fp64"] = accuracy_vs_fp64(dev)
                except Exception as e:
                    import traceback
                    traceback.print_exc()
                    out["alt_math"]["accuracy_vs_fp64"] = {"error": f"{type(e).__name__}: {e}"}
        if train is not None:
            out["train"] = train
            if "value" in train and "value" in legs.get("train_driver", {}):
                legs["train_driver"]["frac_of_train_leg"] = legs["train_driver"]["value"] / train["value"]
        out.update(legs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
