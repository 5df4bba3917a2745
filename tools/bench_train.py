#!/usr/bin/env python3
"""Training throughput (BASELINE config 4 shape): DiT train step = frozen LA-VAE encode -> q_sample ->
forward -> MSE -> backward -> (all-reduce) -> fused AdamW, fp32, synthetic data, per-GPU batch B.

    python tools/bench_train.py --batch 1152 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py ...
"""
import argparse, json, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from t2ms_amd import dist as tdist, synth
from t2ms_amd.train import T2SAdamW, allreduce_gradients


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1152)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--length", type=int, default=96)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--cache_latents", action="store_true", help="encode once (t2ms_amd/latent_cache.py) instead of per step")
    a = ap.parse_args()
    rank, local_rank, world = tdist.env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = tdist.init("nccl", dev)
    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    from model.backbone.DDPM import DDPM
    m = Transformer(); m.load_state_dict(synth.make_dit_state_dict(2025), strict=True); m = m.to(dev).train().set_train_dtype(a.dtype)
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    v.load_state_dict(synth.make_vae_state_dict(2025), strict=True); v = v.to(dev).eval()
    m.encoder = v.encoder
    for n, p in m.named_parameters():
        if "encoder" in n: p.requires_grad = False
    opt = T2SAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    B = a.batch
    x = synth.make_series(1, B, a.length).to(dev)
    text = synth.make_text_embeddings(1, B).to(dev)
    t_ar = 0.0
    z_all = None
    if a.cache_latents:
        from t2ms_amd import latent_cache
        z_all = latent_cache.encode_all(m.encoder, x, dev)
        rows = torch.arange(B, device=dev)

    def step():
        nonlocal t_ar
        if z_all is not None:
            z = z_all[rows]                       # the gather a cached training step does
        else:
            with torch.no_grad():
                z, _ = m.encoder(x)
        t = torch.floor(torch.rand(B, device=dev) * 100).long()
        eps = torch.randn_like(z)
        xt, _ = ddpm.q_sample(z, t, eps)
        opt.zero_grad()
        loss = ddpm.loss(m(input=xt, t=t, text_input=text), eps)
        loss.backward()
        if dist is not None:
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            allreduce_gradients(m, dist)
            torch.cuda.synchronize(dev); t_ar += time.perf_counter() - t0
        opt.step()
        return loss

    for _ in range(a.warmup): step()
    t_ar = 0.0
    tdist.barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(a.steps): loss = step()
    tdist.barrier(dist, dev)
    el = tdist.max_over_ranks(dist, time.perf_counter() - t0, dev)
    if rank == 0:
        flops = 3 * 0.977e9 * B * world * a.steps
        print(json.dumps({"metric": f"DiT training samples/sec (config 4 shape, {a.dtype})", "value": B * world * a.steps / el,
                          "unit": "samples/s", "n_gpus": world, "ms_per_step": el / a.steps * 1e3, "per_gpu_batch": B,
                          "dtype": a.dtype, "latents": "cached" if a.cache_latents else "encoded per step", "tflops": flops / el / 1e12, "allreduce_share": t_ar / el,
                          "loss": float(loss.item())}))
    tdist.barrier(dist, dev)


if __name__ == "__main__":
    main()
