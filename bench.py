#!/usr/bin/env python3
"""Headline benchmark: generated series / second for 1000-step CFG DDPM sampling with the DiT
denoiser (BASELINE.json configs[1]: B=256 per GPU, L=96, cfg 9.0, fp32), synthetic inputs.

One "step" = one pass of the hot path over one batch: x_T (Philox, on device) -> 1000 x
[512-sequence DiT forward + CFG combine + DDPM update] -> LA-VAE decode to (256,96).  Inputs
(text embeddings, weights) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU = weak scaling: every rank samples its own 256-series shard (global rows
[256*rank, 256*rank+256) of the Philox stream); no data-path collective, RCCL only for the
barrier and the max-over-ranks time.
"""
import argparse
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
        out = (C.c_double * 6)()
        L.check(lib.t2s_dit_timing_end(h, out))
    return {"attn_us": out[0] / out[1] * 1e3, "attn_calls": int(out[1]),
            "rows_us": out[2] / out[3] * 1e3, "rows_calls": int(out[3]),
            "other_us": out[4] / out[5] * 1e3, "other_calls": int(out[5]),
            "forward_us": (out[0] + out[2] + out[4]) / n_steps * 1e3}


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


def cpu_baseline(batch, diff_steps, cfg, length, n_cfg_steps=4):
    """The CPU oracle (torch fp32, all host threads) on a bounded sample of the same workload:
    n_cfg_steps CFG steps at the full batch + one decode, extrapolated to diff_steps steps (every
    step costs the same)."""
    from oracle import t2s_oracle as O
    from t2ms_amd import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    sd = synth.make_dit_state_dict(2025)
    vsd = synth.make_vae_state_dict(2025)
    x = synth.make_latents(2025, batch)
    text = synth.make_text_embeddings(2025, batch)
    tab = O.ddpm_tables(diff_steps)
    g = torch.Generator().manual_seed(0)

    def one_step(x, j):
        t = torch.full((batch,), diff_steps - 1 - j, dtype=torch.long)
        u = O.dit_forward(sd, x, t, None)
        c = O.dit_forward(sd, x, t, text)
        return O.ddpm_p_sample(tab, x, u + cfg * (c - u), t, torch.randn(x.shape, generator=g))

    with torch.no_grad():
        x = one_step(x, 0)  # warm
        t0 = time.perf_counter()
        for j in range(1, 1 + n_cfg_steps):
            x = one_step(x, j)
        t_step = (time.perf_counter() - t0) / n_cfg_steps
        t0 = time.perf_counter()
        O.vae_decode(vsd, x, length)
        t_dec = time.perf_counter() - t0
    total = t_step * diff_steps + t_dec
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": batch / total, "unit": "series/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_cfg_steps} CFG steps (2 DiT forwards + DDPM update) at B={batch} + 1 decode, "
                      f"extrapolated x{diff_steps}/{n_cfg_steps}; {t_step:.3f} s/step; CPU: {cpu_model}"}


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
    finally:
        model.set_math("f32")
    return {"math": "bf16x3: every product of the attention and the row chain as six bf16 MFMAs, fp32-accurate (include/t2s.h T2S_MATH_BF16X3)",
            "value": args.batch / el, "unit": "series/s", "ms_per_step": el * 1e3,
            "attention_us": kt["attn_us"], "row_chain_us": kt["rows_us"],
            "attention_bf16_tflops_executed": 6 * FLOP_ATTN_PER_SEQ_BLOCK * 2 * args.batch / (kt["attn_us"] * 1e-6) / 1e12,
            "bf16_dense_peak_tflops": 2500.0}


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
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2],
                    help="sampler lanes: 0 = the library's default (two half-batch chains on two streams from B >= 128), 1, 2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-math", action="store_true", help="skip the extra bf16x3 measurement reported as alt_math")
    ap.add_argument("--math", default="f32", choices=["f32", "bf16x3"],
                    help="matrix arithmetic: f32 MFMA (headline) or fp32-accurate split-bf16 products (include/t2s.h)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from t2ms_amd import dist as tdist
    from t2ms_amd import synth
    from t2ms_amd.sampler import Sampler
    dist = tdist.init("nccl", dev)      # RCCL; None when single-process

    B = args.batch
    model, vae = build_models(dev)
    model.set_math(args.math)
    sampler = Sampler(model, vae.decoder, args.backbone, args.diffusion_steps, args.cfg_scale, B, args.length,
                      dev, use_graph=not args.no_graph, seed=2025, row0=rank * B, lanes=args.lanes)
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
    lanes_used = args.lanes or (int(os.environ["T2S_SAMPLER_LANES"]) if os.environ.get("T2S_SAMPLER_LANES") in ("1", "2")
                                else (2 if B >= 128 and B % 64 == 0 else 1))
    value = total_series / elapsed
    out = {
        "metric": "generated series/sec (B=256, L=96, 1000-step DDPM)",
        "value": value, "unit": "series/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{'configs[1]' if (args.backbone, args.diffusion_steps, B) == ('ddpm', 1000, 256) else 'custom'}: DiT denoiser, {args.diffusion_steps}-step {args.backbone} with CFG "
                               f"(2 forwards/step), B={B}/GPU, L={args.length}, cfg_scale={args.cfg_scale}, "
                               f"LA-VAE decode; Philox noise on device; hipGraph={'off' if args.no_graph else 'on'}; "
                               f"sampler lanes={lanes_used} (half-batch chains on own streams, include/t2s.h t2s_sampler_set_lanes)",
                   "global_batch": B * world, "diffusion_steps": args.diffusion_steps, "parallelism": f"batch-shard x{world}",
                   "sampler_lanes": lanes_used},
    }
    if rank == 0:
        # roofline of the dominant kernel (fused attention; 48 % of all FLOPs), same shapes as the workload
        kt = time_kernels_in_situ(model, dev, lat.clone(), text)
        t_attn = kt["attn_us"] * 1e-6
        flop_attn = FLOP_ATTN_PER_SEQ_BLOCK * 2 * B
        achieved = flop_attn / t_attn / 1e12
        traffic = None
        tfile = os.path.join(REPO, "profiles", "attn_traffic.json")
        if os.path.exists(tfile) and B == 256:     # the PMC passes were taken at the headline shape
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": "attn_fwd_persistent_kernel", "achieved": achieved,
                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                           "traffic": traffic, "hbm_gbps": (traffic / t_attn / 1e9) if traffic else None,
                           "mfma_busy_frac_pmc": 0.79, "pmc_source": "profiles/r01_v7_pmc_mfma_utilisation.md",
                           "avg_launch_us": t_attn * 1e6,
                           "flop_per_launch": flop_attn, "launches_timed": kt["attn_calls"],
                           "timing": "HIP events on the launch stream around every attention launch of "
                                     "8 eager 512-sequence CFG forwards (in situ), the kernel alone on the chip -- the "
                                     "launch shape of `--lanes 1`, whose rocprofv3 average agrees "
                                     "(profiles/*_kernel_stats_lanes1.csv). With 2 lanes the timed region issues this kernel "
                                     "as two 256-sequence launches that time-share the CUs with the other lane's kernels; "
                                     "per-launch durations there are not a kernel property, the pipelined figure is "
                                     "whole_path_frac_of_fp32_mfma_peak"}
        out["kernel_breakdown_us"] = {"attention_x4": kt["attn_us"], "row_chain_x5": kt["rows_us"],
                                      "other_x4": kt["other_us"], "forward_total": kt["forward_us"]}
        # whole-step figure for context: all DiT FLOPs / wall time
        step_flops = FLOP_FORWARD_PER_SEQ * 2 * B * args.diffusion_steps * args.steps
        out["whole_path_tflops"] = step_flops / elapsed / 1e12
        out["whole_path_frac_of_fp32_mfma_peak"] = out["whole_path_tflops"] / PEAK_FP32_MFMA_TFLOPS
        if world == 1 and args.math == "f32" and not args.no_alt_math:
            out["alt_math"] = alt_math_run(model, vae, args, dev, text)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, args.diffusion_steps, args.cfg_scale, args.length)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
