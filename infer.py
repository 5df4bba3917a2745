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
  * loader batches are coalesced: the reference samples one loader batch per launch (default
    `--batch_size 2`, a 4-sequence CFG pass); here the same rows in the same order go `--launch_batch`
    (256) per GPU at a time -- rows are independent and the kernels batch-invariant bit for bit, so
    the files do not change by a byte (`--launch_batch 0` = the reference's launch shape).  The test
    split, its embeddings and every output stay in HBM; the host touches no row inside the loop;
  * under torchrun every rank samples rows [lo,hi) of each launch and keeps them in HBM; the ranks
    meet in ONE gather at the end (a rank without rows joins empty-handed), rank 0 writes;
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

from datafactory.dataloader import epoch_index_batches, loader_provider, resident_tables, walk_index_batches   # noqa: E402
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
        # BASELINE configs[0] (plumbing): the MLP denoiser on the (B,64,L/4) latent, L = 24 (t2s_mlp_forward under no_grad)
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
    identity.  Every step is HIP kernels: the two `model(...)` calls are one launch of t2s_mlp_forward each (row a20), then
    the DDPM update; encoder and decoder likewise; x_T and the per-step draws come from the library's Philox stream keyed
    by the global row."""
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


def launch_plan(n_rows, loader_batch, launch_batch, world):
    """Global row ranges [(s0, s1)] of the sampler launches.  The reference launches once per loader batch
    (infer.py:66-95; default --batch_size 2 = a 4-sequence CFG pass); rows are independent through the whole loop and the
    kernels are batch-invariant bit for bit, so the SAME rows in the SAME order may be sampled `launch_batch` per GPU at a
    time (default 256: the chip-filling shape) without changing a byte of the output.  launch_batch = 0 keeps the
    reference's launch shape (one launch per loader batch)."""
    per = loader_batch if launch_batch <= 0 else launch_batch * world
    return [(s0, min(s0 + per, n_rows)) for s0 in range(0, n_rows, per)]


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
    is_mlp = args.denoiser == "MLP"
    if not is_mlp:
        from t2ms_amd.sampler import default_math
        args.math = getattr(args, "math", None) or default_math()
        model.set_math(args.math)
        if rank == 0:
            print(f"matrix arithmetic: {args.math}" + (" (fp32-accurate split-bf16 products; --math f32 = exact f32 MFMA)" if args.math == "bf16x3" else ""))

    # The loader's ORDER without the loader's per-row work (datafactory.epoch_index_batches draws what one pass over
    # the DataLoader draws): output row i is dataset row order[i], exactly the concatenation of the reference's batches
    # (shuffle=True, drop_last=True, infer.py:66; N = floor(test / B) * B rows).  The test split lives in HBM.
    # (--loader_batches: the same order from a real DataLoader walk over the row numbers -- public torch API only)
    batches = walk_index_batches(dataloader) if getattr(args, "loader_batches", False) else epoch_index_batches(dataloader)
    if rank == 0:
        print("dataset length:", batches.shape[0])
    if batches.shape[0] == 0:
        raise RuntimeError("the test loader produced no full batch (drop_last=True): lower --batch_size")
    order = batches.reshape(-1)
    n_rows, B = int(order.numel()), int(batches.shape[1])
    (series_tab, emb_tab, _), = resident_tables(dataset)
    x1_host = torch.as_tensor(series_tab)[order].float()              # (N, L) fp32: what `x_1.float()` gives per batch
    L = int(x1_host.shape[1])
    x1_dev = x1_host.to(device)
    emb_dev = torch.as_tensor(emb_tab)[order].float().to(device)
    # MLP (configs[0] plumbing): one sampling loop per loader batch, as the reference; DiT: coalesced launches
    plan = launch_plan(n_rows, B, 0 if is_mlp else int(getattr(args, "launch_batch", 256)), world)
    flush_rows = int(os.environ.get("T2S_INFER_FLUSH_ROWS", str(1 << 17)))     # outputs stay in HBM this long (16 KB / row)

    out = None
    if rank == 0:
        out = (np.empty((n_rows, L), np.float32), np.empty((n_rows, 64, 30), np.float32),
               np.empty((n_rows, 64, 30), np.float32))
    held, held_chunks, trace, samplers = [], [], None, {}
    t_start = time.time()

    def flush():
        """Collective: every rank hands over the rows it sampled since the last flush (ONE all_gather of the packed
        (rows, L + 2 * 1920) tensor; a rank without rows joins with an empty tensor) and rank 0 files them by global row."""
        if not held_chunks:
            return
        counts = [sum(tdist.shard_rows(s1 - s0, r, world)[1] - tdist.shard_rows(s1 - s0, r, world)[0]
                      for s0, s1 in held_chunks) for r in range(world)]
        mine = torch.cat(held, dim=0) if held else torch.empty(0, L + 2 * 1920, device=device)
        parts = tdist.gather_ragged(dist, mine, counts, rank)
        if rank == 0:
            for r, part in enumerate(parts):
                if counts[r] == 0:
                    continue
                dest = torch.cat([torch.arange(s0 + tdist.shard_rows(s1 - s0, r, world)[0],
                                               s0 + tdist.shard_rows(s1 - s0, r, world)[1]) for s0, s1 in held_chunks])
                p = part.cpu().numpy()
                d = dest.numpy()
                out[0][d] = p[:, :L]
                out[1][d] = p[:, L:L + 1920].reshape(-1, 64, 30)
                out[2][d] = p[:, L + 1920:].reshape(-1, 64, 30)
        held.clear()
        held_chunks.clear()

    with torch.no_grad():
        for k, (s0, s1) in enumerate(plan):
            lo, hi = tdist.shard_rows(s1 - s0, rank, world)
            n = hi - lo
            if n > 0:                     # a rank without rows in this launch (fewer rows than GPUs) only joins the flush
                x_1, embedding = x1_dev[s0 + lo:s0 + hi], emb_dev[s0 + lo:s0 + hi]
                if is_mlp:
                    lat, series, z_enc = sample_mlp_config1(model, vae, backbone, x_1, embedding, args, device, s0 + lo)
                    lat = torch.nn.functional.pad(lat, (0, 30 - lat.shape[2]))      # the .npy layout is (N,64,30): zero-padded
                else:
                    z_enc, _ = model.encoder(x_1.contiguous())                      # infer.py:73-74
                    sampler = samplers.get(n)
                    if sampler is None:
                        sampler = samplers[n] = Sampler(model, vae.decoder, backbone, args.total_step, args.cfg_scale, n, L,
                                                        device, use_graph=True, seed=args.seed, row0=0)
                    sampler.set_row0(s0 + lo)     # global row index of this shard's first series; the graph is kept
                    want_trace = bool(args.trace) and k == 0 and rank == 0
                    lat, series, tr = sampler.run(embedding.contiguous(), decode=True, trace=want_trace)
                    if want_trace:
                        trace = tr.cpu().numpy()
                held.append(torch.cat([series.reshape(n, L), lat.reshape(n, 1920), z_enc.reshape(n, 1920)], dim=1))
            held_chunks.append((s0, s1))
            if rank == 0:
                print(f"Generating {k}th Batch TS...  (rows {s0}..{s1 - 1} of {n_rows}, {s1 - s0} series per launch)")
            if sum(b - a for a, b in held_chunks) >= flush_rows:
                flush()
        flush()
    tdist.barrier(dist, device)
    loop_s = time.time() - t_start
    args.stats = {"series": n_rows, "loop_s": loop_s, "launches": len(plan), "loader_batch": B, "gpus": world,
                  "series_per_launch_and_gpu": (plan[0][1] - plan[0][0]) // world}
    if rank == 0:
        x_1, (x_t, lat_dec, lat_enc) = x1_host.numpy(), out
        print(f"{n_rows} series in {loop_s:.2f} s ({n_rows / loop_s:.1f} series/s)")
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
    p.add_argument("--no_figs", action="store_true", help="skip the ten fig_i.jpg plots (infer.py:157-163)")
    p.add_argument("--launch_batch", type=int, default=256,
                   help="series per GPU and sampler launch: loader batches are coalesced into launches of this size (same "
                        "rows, same order, same bytes in the files); 0 = one launch per loader batch as the reference")
    p.add_argument("--loader_batches", action="store_true",
                   help="take the row order from a real DataLoader walk (public torch API) instead of the emulated draws of "
                        "datafactory.epoch_index_batches -- same order, same files; the cross-check after a torch upgrade")
    p.add_argument("--math", default=None, choices=["f32", "bf16x3"],
                   help="matrix arithmetic of the DiT: bf16x3 (default; fp32-ACCURATE split-bf16 products on the bf16 matrix cores, "
                        "+35 %%; its error against fp64 is not larger than the reference's PyTorch-CPU fp32 arithmetic, "
                        "profiles/r05_accuracy.md) or f32 (exact f32 MFMA)")
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
    if out is not None and not args.no_figs:
        _save_figs(args.generation_save_path, out[0], out[1])
    return args


if __name__ == "__main__":
    main()
