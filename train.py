#!/usr/bin/env python3
"""Drop-in DiT training driver (reference train.py): same flags, optimizer / scheduler settings,
checkpoint dict and paths -- with the step of train.py:103-127 on the HIP kernels and data
parallelism over GPUs (one process per GPU, ONE flat-bucket RCCL all-reduce per step).

    python train.py --dataset_name ETTh1 --backbone ddpm --denoiser DiT --checkpoint_path ""
    python -m torch.distributed.run --nproc-per-node 8 train.py ...

Kept from the reference: AdamW(lr 1e-4, weight_decay 0) + OneCycleLR(max_lr 1e-4, total_steps =
len(loader) * epochs) stepped per outer batch (mix) / per epoch (split) (train.py:37-38,90,131);
frozen LA-VAE encoder grafted as `model.encoder` (train.py:30-33); the 30 % per-batch
classifier-free text drop drawn from the CPU generator (train.py:120-122); checkpoint
dict(model, optimizer, epoch, loss_list) every 1000 epochs and at the end (train.py:132-136).
Additions: `--synthetic N`, `--random_init`, `--seed`, `--bf16` (BASELINE config 4 arithmetic),
`--no_cache_latents` (by default the frozen encoder runs once per dataset row instead of once per
step: t2ms_amd/latent_cache.py); under torchrun each rank trains on its slice of every batch and
gradients are averaged.
"""
import argparse
import os
import sys
import time
import types

import torch
from torch.optim import lr_scheduler

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from datafactory.dataloader import loader_provider, plan_epochs, resident_tables   # noqa: E402
from model.backbone.DDPM import DDPM                          # noqa: E402
from model.backbone.rectified_flow import RectifiedFlow      # noqa: E402
from model.denoiser.transformer import Transformer            # noqa: E402
from t2ms_amd import dist as tdist                            # noqa: E402
from t2ms_amd import synth                                    # noqa: E402
from t2ms_amd.train import T2SAdamW, allreduce_gradients, allreduce_param_grads      # noqa: E402


def _load_vae(args, device):
    if args.random_init:
        from model.pretrained.vqvae import vqvae
        vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                          embedding_dim=64))
        vae.load_state_dict(synth.make_vae_state_dict(args.seed), strict=True)
    else:
        vae = torch.load(args.pretrained_model_path, map_location="cpu", weights_only=False)   # train.py:22
    return vae.float().to(device).eval()


_SIDE = {}


def _h2d(t, device):
    """CPU -> device on a side stream: a pageable copy makes the host wait for everything queued on ITS stream, which on the
    compute stream means a drained GPU at every step boundary; the side stream is empty, the compute stream waits for its event."""
    if torch.device(device).type != "cuda" or os.environ.get("T2S_SYNC_H2D", "0") not in ("", "0"):
        return t.to(device)
    side = _SIDE.get(str(device))
    if side is None:
        side = _SIDE[str(device)] = torch.cuda.Stream(device)
    with torch.cuda.stream(side):
        out = t.to(device)
    cur = torch.cuda.current_stream(device)
    cur.wait_stream(side)
    out.record_stream(cur)
    return out


TIME_KEY = 0x74696D65       # "time": xor-ed into the seed for the per-row diffusion-time stream (the noise stream uses "rain")


def train_step(model, backbone, opt, dist, args, x_1, emb, device, rank, world, latents=None, idx=None, step_no=0,
               emb_table=None, drop_text=None):
    """One optimisation step on this rank's slice of the batch (train.py:103-127 / 60-87).

    Two ways to name the batch: host tensors `x_1` (n, L) / `emb` (n, 128) as the DataLoader collates them, or -- the
    resident path -- `idx` (n,) dataset rows into the device tables `latents` (N, 64, 30) and `emb_table` (N, 128) with
    `x_1 = emb = None`: the rank then takes only ITS slice of the index vector (a device tensor is sliced in place: the
    resident path of train() uploads a whole plan of index batches at a time, so a step issues NO host -> device copy).

    Every rank runs every step -- also with an EMPTY slice (a length group smaller than the world size): it then
    contributes a zero bucket, because the gradient all-reduce is a collective.  All random draws are functions of
    (seed, step, GLOBAL row): the per-row diffusion time t (train.py:109,113: `torch.rand`) and the Gaussian targets come
    from the library's Philox streams keyed by the global row, drawn ON the device (t2s_philox_uniform / _normal: no
    host -> device copy per step); the CFG coin (train.py:120-122) from the CPU generator, identically seeded on every
    rank (`drop_text` hands in a coin train() drew ahead, in the same order).  The step is therefore the same computation
    for any number of GPUs (up to the summation order of the all-reduce), and the gradient is that of the mean loss over
    the GLOBAL batch: rank r's bucket is weighted n_r / n."""
    from t2ms_amd.sampler import philox_normal, philox_uniform
    n_global = int(x_1.shape[0] if x_1 is not None else idx.shape[0])
    lo, hi = tdist.shard_rows(n_global, rank, world)
    n = hi - lo
    if drop_text is None:
        drop_text = bool(torch.rand(1) < 0.3)                  # classifier-free guidance coin (train.py:120-122)
    u = philox_uniform(n, 1, args.seed ^ TIME_KEY, step_no, lo, device).view(n) if n > 0 else None
    opt.zero_grad()
    loss = None
    is_mlp = getattr(args, "denoiser", "DiT") == "MLP"
    if n > 0 and is_mlp:
        # BASELINE configs[0] in the runnable form SURVEY.md 8(d) prescribes: the MLP denoiser (t2s_mlp_forward / t2s_mlp_backward behind its autograd node) diffuses
        # the PRE-interpolation latent `before` (B,64,L/4 = 6); encoder, q_sample and the loss are the HIP kernels
        emb = _h2d(emb[lo:hi].float(), device)
        with torch.no_grad():
            _, z = model.encoder(_h2d(x_1[lo:hi].float(), device).contiguous())
        if z.shape[2] != 6:
            raise ValueError(f"the MLP denoiser needs the 6-wide latent of L = 24 series (mlp.py:67), got L/4 = {z.shape[2]}")
        if args.backbone != "ddpm":
            raise ValueError("config 1 (MLP denoiser) is wired for --backbone ddpm")
        noise = philox_normal(n, z[0].numel(), args.seed ^ 0x7261696E, step_no, lo, device).view_as(z)
        t = torch.floor(u * args.total_step).long()
        x_t, _ = backbone.q_sample(z.contiguous(), t, noise)
        pred = model(x_t, t, None if drop_text else emb)
        loss = backbone.loss(pred, noise)
        loss.backward()
    elif n > 0:
        enc_trains = any(p.requires_grad for p in model.encoder.parameters())
        idx_dev = None
        if latents is not None and idx is not None:
            idx_dev = idx[lo:hi] if idx.is_cuda else _h2d(idx[lo:hi], device)
        emb = emb_table[idx_dev] if emb_table is not None else _h2d(emb[lo:hi].float(), device)
        if idx_dev is not None:
            z = latents[idx_dev]                                                       # pre-encoded rows (latent cache)
        elif enc_trains:
            z, _ = model.encoder(_h2d(x_1[lo:hi].float(), device).contiguous())        # un-frozen encoder (train.py:31-33): autograd
        else:
            with torch.no_grad():
                z, _ = model.encoder(_h2d(x_1[lo:hi].float(), device).contiguous())    # frozen LA-VAE (train.py:31-33,106)
        noise = philox_normal(n, z[0].numel(), args.seed ^ 0x7261696E, step_no, lo, device).view_as(z)
        if args.backbone == "flowmatching":
            t = torch.round(u * args.total_step) / args.total_step                     # train.py:109
            if enc_trains:       # the same two formulas as differentiable torch glue (rectified_flow.py:8-12)
                tt = t.float()[:, None, None]
                x_t, x_0 = tt * z + (1 - tt) * noise, noise
            else:
                x_t, x_0 = backbone.create_flow(z, t, x_0=noise)
            target = z - x_0
        elif args.backbone == "ddpm":
            t = torch.floor(u * args.total_step).long()                                # train.py:113
            target = noise
            if enc_trains:       # DDPM.py:19-27 as differentiable torch glue
                ab = backbone.alpha_bar.gather(-1, t).reshape(-1, 1, 1)
                x_t = ab ** 0.5 * z + (1 - ab) ** 0.5 * noise
            else:
                x_t, _ = backbone.q_sample(z, t, target)
        else:
            raise ValueError(f"Unsupported backbone type: {args.backbone}")
        pred = model(input=x_t, t=t, text_input=None if drop_text else emb)
        loss = backbone.loss(pred, target)
        loss.backward()
    elif args.backbone not in ("flowmatching", "ddpm"):
        raise ValueError(f"Unsupported backbone type: {args.backbone}")
    if dist is not None and is_mlp:
        loss = allreduce_param_grads([p for nm, p in model.named_parameters() if "encoder" not in nm], dist, n_local=n,
                                     n_global=n_global, loss=loss)
    elif dist is not None:
        enc_params = [p for p in model.encoder.parameters() if p.requires_grad]
        if enc_params:           # un-frozen encoder: its gradients travel as a second flat message
            allreduce_param_grads(enc_params, dist, n_local=n, n_global=n_global)
        _, loss = allreduce_gradients(model, dist, n_local=n, n_global=n_global,
                                      loss=loss if loss is not None else torch.zeros((), device=device))
    opt.step()
    return loss


def train(args):
    device = torch.device(args.device)
    rank, _, world = tdist.env_world()
    dist = tdist.init("nccl", device)
    args.seed = tdist.broadcast_int(dist, args.seed)
    if rank == 0:
        print(f"Training config::\tepoch: {args.epochs}\tsave_path: {args.save_path}\tdevice: {args.device}\tGPUs: {world}")
        os.makedirs(args.save_path, exist_ok=True)
    torch.manual_seed(args.seed)              # identical shuffles / t / CFG coin on every rank (CPU generator)
    dataset, dataloader = loader_provider(args, period="train")
    from model.denoiser.mlp import MLP
    model = {"DiT": Transformer, "MLP": MLP}.get(args.denoiser)                 # train.py:16
    if model is None:
        raise ValueError("No denoiser found")
    if args.denoiser == "MLP":
        # config 1 plumbing: the reference's MLP needs a 6-wide latent (mlp.py:55,67), i.e. fixed-length L = 24 data
        if args.mix_train:
            raise ValueError("--denoiser MLP needs fixed-length L = 24 series (its latent is (B,64,6)): use --split_train "
                             "with a *_24 dataset")
        args.cache_latents = False            # the cache holds the interpolated (B,64,30) latents of the DiT path
    model = model()
    if args.random_init:
        sd = synth.make_dit_state_dict(args.seed) if args.denoiser == "DiT" else synth.make_mlp_state_dict(args.seed)
        model.load_state_dict(sd, strict=True)
    model = model.to(device)
    vae = _load_vae(args, device)
    backbone = {"flowmatching": RectifiedFlow(), "ddpm": DDPM(args.total_step, args.device)}.get(args.backbone)
    if backbone is None:
        raise ValueError("No backbone found")
    model.encoder = vae.encoder
    if args.bf16:
        if args.denoiser != "DiT":
            raise ValueError("--bf16 selects the DiT training kernels' arithmetic; the MLP denoiser trains in fp32 (t2s_mlp_backward)")
        model.set_train_dtype("bf16")
    for name, p in model.named_parameters():
        if "encoder" in name:
            p.requires_grad = not args.usepretrainedvae
    if not args.usepretrainedvae:
        # train.py:31-33: the grafted LA-VAE encoder trains jointly with the denoiser.  (As in the reference the flag is an
        # untyped argparse string, so only an EMPTY value -- `--usepretrainedvae ""` -- is false; "False" is a non-empty string.)
        # The encoder then runs as torch ops under autograd (plumbing; model/pretrained/vqvae.py) and the DiT hands back the
        # gradient of its input latent (t2s_dit_train_input_grad); latents cannot be cached.
        if args.denoiser != "DiT":
            raise ValueError("--usepretrainedvae false is wired for the DiT denoiser")
        args.cache_latents = False
    if rank == 0:
        print(f"Total learnable parameters: {sum(p.numel() for p in model.parameters() if p.requires_grad)}")
    # every parameter, as the reference builds it (train.py:37): the optimizer state_dict then indexes the same 67
    # tensors (pos_embed first, the frozen encoder last) and checkpoints interchange; tensors without a gradient
    # (frozen, or the never-used unpatch.*) are skipped by step() and carry no state, as in torch.optim.AdamW
    opt = T2SAdamW(model.parameters(), lr=1e-4, weight_decay=0.0)
    sched = lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=max(1, len(dataloader) * args.epochs))
    loss_list, start_epoch = [], 0
    if args.checkpoint_path:
        ck = torch.load(args.checkpoint_path, map_location=device)
        model.load_state_dict(ck["model"])
        opt.load_state_dict(ck["optimizer"])
        start_epoch, loss_list = ck["epoch"] + 1, ck["loss_list"]
    cache = None
    from t2ms_amd import latent_cache
    if args.cache_latents:
        cache = latent_cache.attach(dataset, model.encoder, device)
        if rank == 0:
            print("latent cache: " + ", ".join(f"L={L}: {tuple(z.shape)}" for L, z in sorted(cache.items())))
    # resident path (default whenever the latent cache is attached): the epoch's index batches come from the seeded
    # generator exactly as a pass over the DataLoader draws them (datafactory.epoch_index_batches), rows are gathered from
    # device tables, and each rank touches only its slice of every index vector -- no per-row __getitem__ / collate hop
    # (73 ms of host time per 9,216-row batch and rank against 11 ms of kernels on 8 GPUs).  --loader_batches walks the
    # DataLoader itself, as the reference does (train.py:101-131); both orders and both loss lists are identical.
    resident = cache is not None and not getattr(args, "loader_batches", False)
    if resident:
        tabs = resident_tables(dataset)
        leaves = latent_cache.leaf_datasets(dataset)
        lat_tabs = [d.latents for d in leaves]
        emb_tabs = [torch.as_tensor(e).float().to(device) for _, e, _ in tabs]
        starts = [st for _, _, st in tabs]
    model.train()
    t0, seen = time.time(), 0
    step_no = len(loss_list)                  # global optimisation-step counter (keys the noise stream; survives resume)
    pending = []                              # losses of the steps since the last flush, still on the device
    max_steps = int(getattr(args, "max_steps", 0) or 0)
    on_step = getattr(args, "on_step", None)  # bench.py's clock: called after every optimisation step is ENQUEUED
    steps_done, stop = 0, False

    def flush():
        if pending:
            loss_list.extend(torch.stack(pending).tolist())
            pending.clear()

    # Resident path: a PLAN of index batches for several epochs at once.  Per epoch the CPU generator is consumed exactly
    # as the loader pass + the steps' CFG coins consume it (two draws when the pass starts, then one coin per length group in
    # visiting order), the rows of a batch are ordered by length group (stable: the collate's grouping, dataloader.py:115-133)
    # and made local to their dataset; the plan's index array goes to the device in ONE copy -- every >= PLAN_STEPS
    # optimisation steps even for a set with one batch per epoch -- and a step slices it in place.
    PLAN_STEPS = 64

    def epoch_batches(first_epoch):
        """-> (epoch, batch index, [(x_1, emb, latents, idx, emb_table, coin)]) for every loader batch from `first_epoch` on."""
        if not resident:
            for e in range(first_epoch, args.epochs):
                for b, data in enumerate(dataloader):
                    groups = data if args.mix_train else [data]
                    yield e, b, [(g[1], g[2], cache.get(int(g[1].shape[1])) if cache else None, g[3] if len(g) > 3 else None,
                                  None, None) for g in groups]
            return
        e = first_epoch
        while e < args.epochs:
            plans, flat = plan_epochs(dataloader, starts, args.mix_train, e, args.epochs, PLAN_STEPS)
            flat_dev = _h2d(flat, device)
            off = 0
            for ep, rows, groups, coins in plans:
                for b in range(rows.shape[0]):
                    out, o = [], off + b * rows.shape[1]
                    for (w, c), coin in zip(groups[b], coins[b]):
                        out.append((None, None, lat_tabs[w], flat_dev[o:o + c], emb_tabs[w], coin))
                        o += c
                    yield ep, b, out
                off += rows.numel()
            e += len(plans)

    def end_of_epoch(epoch):
        if not args.mix_train:
            sched.step()
        if (epoch % 1000 == 0 or epoch == args.epochs - 1) and rank == 0:
            flush()
            print(f"Saving model {epoch} to {args.save_path}...")
            torch.save(dict(model=model.state_dict(), optimizer=opt.state_dict(), epoch=epoch, loss_list=loss_list),
                       os.path.join(args.save_path, f"model_{epoch}.pth"))

    epoch = None
    for ep, batch, groups in epoch_batches(start_epoch):
        if epoch is not None and ep != epoch:
            end_of_epoch(epoch)
        epoch = ep
        for x_1, emb, lat, idx, emb_tab, coin in groups:
            loss = train_step(model, backbone, opt, dist, args, x_1, emb, device, rank, world, lat, idx, step_no, emb_tab, coin)
            n_rows = int(x_1.shape[0] if x_1 is not None else idx.shape[0])
            step_no += 1
            steps_done += 1
            seen += n_rows
            pending.append(loss.detach())           # .item() here (train.py:126) would drain the GPU at every step
            if on_step is not None:
                on_step(steps_done, n_rows, model)
            # the reference prints at batch 0 of EVERY epoch (train.py:128-129): with the 1-2 batches per epoch of an
            # ETTh1-sized set that would be a device sync per step -- report on a step count instead
            if steps_done == 1 or steps_done % 100 == 0:
                flush()
                if rank == 0:
                    print(f"[Epoch {epoch}] [batch {batch}] loss: {loss_list[-1]:.6f}  "
                          f"({seen / (time.time() - t0):.1f} samples/s)")
            if max_steps and steps_done >= max_steps:
                stop = True
                break
        if stop:
            break
        if args.mix_train:
            sched.step()
    if epoch is not None and not stop:
        end_of_epoch(epoch)
    flush()
    tdist.barrier(dist, device)
    return loss_list


def get_args(argv=None):
    p = argparse.ArgumentParser(description="Train T2S model")
    p.add_argument("--checkpoint_path", type=str,
                   default="./results/denoiser_results/checkpoints/flowmatching_DiT_weather/model_6000.pth")
    p.add_argument("--dataset_name", type=str, default="weather", help="dataset name")
    p.add_argument("--batch_size", type=int, default=9216, help="batch_size")
    p.add_argument("--epochs", type=int, default=20000, help="training epochs")
    p.add_argument("--save_path", type=str, default="./results/denoiser_results", help="denoiser model save path")
    p.add_argument("--mix_train", type=bool, default=True, help="mixture train or not")
    p.add_argument("--usepretrainedvae", default=True, help="pretrained vae")
    p.add_argument("--total_step", type=int, default=100, help="sampling from [0,1]")
    p.add_argument("--backbone", type=str, default="flowmatching", help="flowmatching or ddpm or edm")
    p.add_argument("--denoiser", type=str, default="DiT", help="DiT or MLP")
    p.add_argument("--seed", type=int, default=2025)
    p.add_argument("--synthetic", type=int, default=0, help="serve N synthetic rows per length instead of the CSVs")
    p.add_argument("--random_init", action="store_true", help="seeded synthetic LA-VAE / DiT weights")
    p.add_argument("--split_train", action="store_true", help="mix_train=False (argparse type=bool cannot be switched off)")
    p.add_argument("--bf16", action="store_true", help="bf16 MFMA operands / saved activations (fp32 accumulate and master weights)")
    p.add_argument("--no_cache_latents", dest="cache_latents", action="store_false",
                   help="re-run the frozen LA-VAE encoder on every step, as the reference does")
    p.add_argument("--loader_batches", action="store_true",
                   help="walk the DataLoader row by row as the reference does instead of gathering index batches from the "
                        "device-resident tables (same order, same losses)")
    p.add_argument("--max_steps", type=int, default=0, help="stop after this many optimisation steps (0: run all epochs)")
    args = p.parse_args(argv)
    if args.split_train:
        args.mix_train = False
    if not torch.cuda.is_available():
        sys.exit("train.py: no GPU visible -- this build runs the HIP path only (no CPU fallback)")
    local_rank = tdist.local_device_index()
    torch.cuda.set_device(local_rank)
    args.device = f"cuda:{local_rank}"
    if args.mix_train:
        args.data_length = 0
    root = args.dataset_name.split("_")[0]
    args.pretrained_model_path = f"results/saved_pretrained_models/dataset{root}_epoch2000/final_model.pth"
    args.save_path = os.path.join(args.save_path, "checkpoints",
                                  "{}_{}_{}".format(args.backbone, args.denoiser, args.dataset_name))
    return args


if __name__ == "__main__":
    a = get_args()
    t_start = time.time()
    train(a)
    print(time.time() - t_start)
