#!/usr/bin/env python3
"""Drop-in sampling driver (reference infer.py): same flags, same checkpoint / output paths, same
four .npy files -- with the loop of infer.py:76-95 run by the fused HIP sampler.

    python infer.py --dataset_name ETTh1_96 --backbone ddpm --denoiser DiT --total_step 1000 --cfg_scale 9
    python -m torch.distributed.run --nproc-per-node 8 infer.py ...      # batch sharded over GPUs

Differences from the reference, all additive:
  * `--seed` (the reference never seeds, infer.py:15-25 is dead code) keys the on-device Philox
    noise stream; results do not depend on the number of GPUs;
  * `--synthetic N` / `--random_init` let the driver run without the (offline-unavailable) CSVs and
    trained checkpoints;
  * under torchrun every rank samples rows [lo,hi) of each batch, rank 0 gathers and writes;
  * the per-step decode of the first batch (infer.py:90-93, GIF only) is `--trace`, off by default.
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from datafactory.dataloader import loader_provider            # noqa: E402
from model.backbone.DDPM import DDPM                          # noqa: E402,F401  (API parity)
from model.backbone.rectified_flow import RectifiedFlow      # noqa: E402,F401
from model.denoiser.transformer import Transformer            # noqa: E402
from t2ms_amd import dist as tdist                            # noqa: E402
from t2ms_amd import synth                                    # noqa: E402
from t2ms_amd.sampler import Sampler, np_save_outputs         # noqa: E402


def _weight_seed(args):
    """--random_init: the synthetic weights keep the seed the driver started with (--run_multi advances the NOISE seed)."""
    return getattr(args, "weight_seed", args.seed)


def _load_models(args, device):
    root = args.dataset_name.split("_")[0]
    if args.random_init:
        from model.pretrained.vqvae import vqvae
        vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                          embedding_dim=64))
        vae.load_state_dict(synth.make_vae_state_dict(_weight_seed(args)), strict=True)
    else:
        vae = torch.load(f"results/saved_pretrained_models/dataset{root}_epoch2000/final_model.pth",
                         map_location=torch.device("cpu"), weights_only=False)     # infer.py:39
    vae = vae.float().to(device).eval()
    if args.denoiser == "MLP":
        # BASELINE configs[0] (plumbing): torch mirror of the MLP denoiser on the (B,64,L/4) latent, L = 24
        from model.denoiser.mlp import MLP
        model = MLP()
        if args.random_init:
            model.load_state_dict(synth.make_mlp_state_dict(_weight_seed(args)), strict=True)
        else:
            model.load_state_dict(torch.load(args.checkpoint_path, map_location="cpu")["model"], strict=False)
        model.encoder = vae.encoder
        return model.to(device).eval(), vae
    if args.denoiser != "DiT":
        raise ValueError("No denoiser found")
    model = Transformer().to(device)
    model.encoder = vae.encoder                                                     # infer.py:47
    if args.random_init:
        sd = synth.make_dit_state_dict(_weight_seed(args))
        sd.update({"encoder." + k: v for k, v in vae.encoder.state_dict().items()})
        model.load_state_dict(sd, strict=True)
    else:
        model.load_state_dict(torch.load(args.checkpoint_path, map_location="cpu")["model"])   # infer.py:48
    return model.to(device).eval(), vae


def sample_mlp_config1(model, vae, backbone, x_1, embedding, args, device, row0):
    """The loop of infer.py:76-95 for `--denoiser MLP`, in the runnable form SURVEY.md 8(d) config 1 prescribes: the
    reference's MLP needs a 6-wide latent (mlp.py:55,67) while its encoder emits 30 (vqvae.py:70), so the diffusion
    state is the PRE-interpolation latent `before` (B,64,L/4), L = 24, and the decoder's 6 -> 6 interpolation is the
    identity.  The MLP is a torch module (plumbing); the encoder, the DDPM update and the decoder are
    the HIP kernels; x_T and the per-step draws come from the library's Philox stream keyed by the global row."""
    from t2ms_amd.sampler import XT_STREAM, philox_normal
    if backbone != "ddpm":
        raise ValueError("config 1 (MLP denoiser) is wired for --backbone ddpm")
    z_enc, before = model.encoder(x_1.contiguous())
    B, C, W = before.shape
    if W != 6:
        raise ValueError(f"the MLP denoiser needs the 6-wide latent of L = 24 series (mlp.py:67), got L/4 = {W}")
    from model.backbone.DDPM import DDPM as _DDPM
    ddpm = _DDPM(args.total_step, device)
    x_t = philox_normal(B, C * W, args.seed, XT_STREAM, row0, device).view(B, C, W)
    for j in range(args.total_step):
        t = torch.full((B,), args.total_step - 1 - j, dtype=torch.long, device=device)
        u = model(x_t, t, None)
        c = model(x_t, t, embedding)
        pred = u + args.cfg_scale * (c - u)                                               # infer.py:87
        x_t = ddpm.p_sample(x_t, pred, t, eps=philox_normal(B, C * W, args.seed, j, row0, device).view(B, C, W))
    series, _ = vae.decoder(x_t, length=x_1.shape[-1])
    return x_t, series.reshape(B, -1), z_enc


def infer(args):
    device = torch.device(args.device)
    rank, local_rank, world = tdist.env_world()
    dist = tdist.init("nccl", device)
    # ONE seed for the whole job: it seeds the loader shuffle (every rank must walk the same batches to take ITS rows
    # of each) and keys the Philox noise.  A per-rank time-based default would silently mis-pair series across ranks.
    args.seed = tdist.broadcast_int(dist, args.seed)
    backbone = {"flowmatching": "flowmatching", "ddpm": "ddpm"}.get(args.backbone)
    if backbone is None:
        raise ValueError("No backbone found")
    if rank == 0:
        print(f"Inference config::Step: {args.total_step}\t CFG Scale: {args.cfg_scale}\t "
              f"Use Pretrained VAE: {args.usepretrainedvae}\t GPUs: {world}")
        os.makedirs(args.generation_save_path_result, exist_ok=True)
    torch.manual_seed(args.seed)          # identical loader shuffle on every rank
    dataset, dataloader = loader_provider(args, period="test")
    model, vae = _load_models(args, device)
    if args.denoiser == "DiT":
        model.set_math(getattr(args, "math", "f32"))

    x1_all, xt_all, lat_dec_all, lat_enc_all, trace = [], [], [], [], None
    sampler, t_start, n_series = None, time.time(), 0
    with torch.no_grad():
        for batch, (y, x_1, embedding) in enumerate(dataloader):
            B, L = x_1.shape[0], x_1.shape[-1]
            lo, hi = tdist.shard_rows(B, rank, world)
            x_1 = x_1.float().to(device)
            embedding = embedding.float().to(device)
            if args.denoiser == "MLP":
                lat, series, z_enc = sample_mlp_config1(model, vae, backbone, x_1[lo:hi], embedding[lo:hi], args, device,
                                                        n_series + lo)
                lat = torch.nn.functional.pad(lat, (0, 30 - lat.shape[2]))      # the .npy layout is (N,64,30): zero-padded
                series = tdist.gather_rows(dist, series, B, rank, world)
                lat = tdist.gather_rows(dist, lat, B, rank, world)
                z_enc = tdist.gather_rows(dist, z_enc, B, rank, world)
                n_series += B
                if rank == 0:
                    print(f"Generating {batch}th Batch TS...  ({n_series / (time.time() - t_start):.1f} series/s)")
                    x1_all.append(x_1.cpu().numpy().squeeze())
                    xt_all.append(series.cpu().numpy().squeeze())
                    lat_dec_all.append(lat.cpu().numpy().squeeze())
                    lat_enc_all.append(z_enc.cpu().numpy().squeeze())
                continue
            z_enc, _ = model.encoder(x_1[lo:hi].contiguous())                      # infer.py:73-74
            if sampler is None or sampler.batch != hi - lo or sampler.length != L:
                sampler = Sampler(model, vae.decoder, backbone, args.total_step, args.cfg_scale, hi - lo, L,
                                  device, use_graph=True, seed=args.seed, row0=0)
            sampler.set_row0(n_series + lo)       # global row index of this shard's first series; the graph is kept
            want_trace = bool(args.trace) and batch == 0 and rank == 0
            lat, series, tr = sampler.run(embedding[lo:hi].contiguous(), decode=True, trace=want_trace)
            if want_trace:
                trace = tr.cpu().numpy()
            series = tdist.gather_rows(dist, series, B, rank, world)
            lat = tdist.gather_rows(dist, lat, B, rank, world)
            z_enc = tdist.gather_rows(dist, z_enc, B, rank, world)
            n_series += B
            if rank == 0:
                print(f"Generating {batch}th Batch TS...  ({n_series / (time.time() - t_start):.1f} series/s)")
                x1_all.append(x_1.cpu().numpy().squeeze())
                xt_all.append(series.cpu().numpy().squeeze())
                lat_dec_all.append(lat.cpu().numpy().squeeze())
                lat_enc_all.append(z_enc.cpu().numpy().squeeze())
    if rank == 0:
        if not x1_all:
            raise RuntimeError("the test loader produced no full batch (drop_last=True): lower --batch_size")
        x_1 = np.concatenate([a.reshape(-1, a.shape[-1]) for a in x1_all], axis=0)
        x_t = np.concatenate([a.reshape(-1, a.shape[-1]) for a in xt_all], axis=0)
        lat_dec = np.concatenate([a.reshape(-1, 64, 30) for a in lat_dec_all], axis=0)
        lat_enc = np.concatenate([a.reshape(-1, 64, 30) for a in lat_enc_all], axis=0)
        np_save_outputs(args.generation_save_path_result, x_1, x_t, lat_dec, lat_enc)   # infer.py:118-123
        if trace is not None:
            np.save(os.path.join(args.generation_save_path_result, "x_infer_trace.npy"), trace)
        print(f"saved {x_1.shape[0]} series to {args.generation_save_path_result}")
    tdist.barrier(dist, device)
    return (x_1[:, :, None], x_t[:, :, None], lat_dec, lat_enc) if rank == 0 else None


def _save_figs(path, x_1, x_t):
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
    except Exception:                                          # plotting is optional
        return
    for i in range(min(10, x_1.shape[0])):
        plt.clf()
        plt.plot(x_1[i], label="ground truth")
        plt.plot(x_t[i], label="generated")
        plt.legend()
        plt.savefig(os.path.join(path, f"fig_{i}.jpg"))


def build_parser():
    p = argparse.ArgumentParser(description="Inference flow matching model")
    p.add_argument("--batch_size", type=int, default=2, help="batch size")
    p.add_argument("--save_path", type=str, default="./results/denoiser_results", help="Denoiser Model save path")
    p.add_argument("--usepretrainedvae", default=True, help="pretrained vae")
    p.add_argument("--backbone", type=str, default="flowmatching", help="flowmatching or DDPM or EDM")
    p.add_argument("--denoiser", type=str, default="DiT", help="DiT or MLP")
    p.add_argument("--cfg_scale", type=float, default=7, help="CFG Scale")
    p.add_argument("--total_step", type=int, default=100, help="total step sampled from [0,1]")
    p.add_argument("--checkpoint_id", type=int, default=19999, help="model id")
    p.add_argument("--dataset_name", type=str, default="exchangerate_24", help="dataset name")
    p.add_argument("--run_multi", type=bool, default=False, help="run multi times for CRPS,MAP,MRR,NDCG")
    # additions (see module docstring)
    p.add_argument("--seed", type=int, default=None, help="Philox noise seed (default: time based)")
    p.add_argument("--synthetic", type=int, default=0, help="serve N synthetic rows instead of the CSV")
    p.add_argument("--random_init", action="store_true", help="seeded synthetic weights instead of checkpoints")
    p.add_argument("--trace", action="store_true", help="decode row 0 after every step of the first batch")
    p.add_argument("--math", default="f32", choices=["f32", "bf16x3"],
                   help="matrix arithmetic of the DiT: f32 MFMA (default) or fp32-accurate split-bf16 products (faster)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    args.mix_train = False
    if not torch.cuda.is_available():
        sys.exit("infer.py: no GPU visible -- this build runs the HIP path only (no CPU fallback)")
    local_rank = tdist.local_device_index()
    torch.cuda.set_device(local_rank)
    args.device = f"cuda:{local_rank}"
    if args.seed is None:
        args.seed = int(time.time()) & 0x7FFFFFFF
    args.weight_seed = args.seed
    root = args.dataset_name.split("_")[0]
    args.checkpoint_path = os.path.join(args.save_path, "checkpoints", f"{args.backbone}_{args.denoiser}_{root}",
                                        f"model_{args.checkpoint_id}.pth")
    args.generation_save_path = os.path.join(
        args.save_path, "generation",
        "{}_{}_{}_{}_{}".format(args.backbone, args.denoiser, args.dataset_name, args.cfg_scale, args.total_step))
    print("start generate", args.run_multi)
    args.generation_save_path_result = args.generation_save_path
    out = infer(args)
    if args.run_multi:                                           # infer.py:148-164: 1 + 10 runs
        for run_index in range(10):
            args.generation_save_path_result = os.path.join(args.generation_save_path, f"run_{run_index}")
            args.seed += 1
            out = infer(args)
    if out is not None:
        _save_figs(args.generation_save_path, out[0], out[1])


if __name__ == "__main__":
    main()
