"""RCCL itself, on one rank.  The two-rank rehearsals (tests/test_two_ranks_one_gpu.py) must swap in gloo
because RCCL refuses two ranks on one device, which leaves `init_process_group("nccl", device_id=...)`, the device-side
all-reduce of the flat gradient bucket, all_gather on device tensors, the object broadcast and barrier() under NCCL
unexecuted before the driver's 8-GPU run.  A ONE-rank NCCL communicator is legal: T2S_FORCE_DIST=1 makes
t2ms_amd.dist.init build the process group at world size 1, and the same code paths then run through librccl on the test
GPU.  Every child is a fresh process that initialises the process group before any other GPU call.

Checked: librccl is mapped into the child; results through the collectives equal the `dist = None` run bit for bit
(one rank: a SUM all-reduce is the identity, so any difference is a bug in the bucket / weighting / loss-slot code).
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script_and_args, port, force, cwd, timeout=420):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "T2S_DIST_BACKEND", "T2S_SHARE_GPU", "T2S_FORCE_DIST"):
        env.pop(k, None)
    if force:
        env["T2S_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable] + script_and_args, env=env, cwd=cwd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{script_and_args}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r


WORKER = textwrap.dedent("""
    import json, os, sys, types
    import torch
    from t2ms_amd import dist as tdist
    rank, _, world = tdist.env_world()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist = tdist.init("nccl", dev)                      # first GPU-touching call of the process
    forced = os.environ.get("T2S_FORCE_DIST") == "1"
    assert (dist is not None) == forced
    info = {"forced": forced}
    if forced:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import train as T                                   # the drop-in driver at the repo root
    from t2ms_amd import synth
    from t2ms_amd.train import T2SAdamW, N_GRAD, _trainable
    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    from model.backbone.DDPM import DDPM

    # ---- host-side helpers through the communicator
    info["seed"] = tdist.broadcast_int(dist, 424242)
    info["max"] = tdist.max_over_ranks(dist, 1.25, dev)
    rows = torch.arange(7 * 96, dtype=torch.float32, device=dev).reshape(7, 96)
    got = tdist.gather_rows(dist, rows, 7, rank, world)
    assert torch.equal(got, rows)
    tdist.barrier(dist, dev)
    t = torch.full((1000,), 3.0, device=dev)
    tdist.all_reduce_sum(dist, t)
    assert bool((t == 3.0).all())

    # ---- the training step of train.py with the flat-bucket all-reduce (loss rides in the spare slot)
    args = types.SimpleNamespace(backbone="ddpm", total_step=100, seed=2025)
    torch.manual_seed(args.seed)
    m = Transformer(); m.load_state_dict(synth.make_dit_state_dict(2025), strict=True); m = m.to(dev).train()
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    v.load_state_dict(synth.make_vae_state_dict(2025), strict=True); v = v.to(dev).eval()
    m.encoder = v.encoder
    for n, p in m.named_parameters():
        if "encoder" in n: p.requires_grad = False
    opt = T2SAdamW(m.parameters(), lr=1e-4, weight_decay=0.0)
    ddpm = DDPM(100, dev)
    rec = {"flat": [], "loss": []}
    for step_no, B in enumerate((6, 6, 3)):
        x = synth.make_series(10 + step_no, B, 96)
        emb = synth.make_text_embeddings(10 + step_no, B)
        loss = T.train_step(m, ddpm, opt, dist, args, x, emb, dev, rank, world, None, None, step_no)
        rec["flat"].append(m.__dict__["_t2s_flat_grad"][:N_GRAD].detach().cpu().clone())
        rec["loss"].append(float(loss))
    rec["weights"] = torch.cat([p.detach().reshape(-1).cpu() for p in _trainable(m)])
    torch.cuda.synchronize()
    maps = open("/proc/self/maps").read()
    info["rccl_mapped"] = "librccl" in maps
    info["t2s_mapped"] = "libt2s_hip.so" in maps
    torch.save(rec, os.path.join(sys.argv[1], f"rec_{int(forced)}.pt"))
    json.dump(info, open(os.path.join(sys.argv[1], f"info_{int(forced)}.json"), "w"))
    tdist.barrier(dist, dev)
    if dist is not None:
        dist.destroy_process_group()
    print("WORKER OK", forced)
""")


def test_rccl_one_rank_collectives_and_training_step_equal_the_undistributed_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    _run([str(script), str(tmp_path)], 29561, False, REPO)
    _run([str(script), str(tmp_path)], 29562, True, REPO)
    plain = torch.load(tmp_path / "rec_0.pt")
    rccl = torch.load(tmp_path / "rec_1.pt")
    info = json.load(open(tmp_path / "info_1.json"))
    assert info["forced"] and info["rccl_mapped"] and info["t2s_mapped"], info
    assert info["seed"] == 424242 and info["max"] == 1.25
    for step in range(3):
        assert torch.equal(plain["flat"][step], rccl["flat"][step]), f"step {step}: the all-reduced bucket differs"
        assert plain["loss"][step] == rccl["loss"][step], (step, plain["loss"], rccl["loss"])
    assert torch.equal(plain["weights"], rccl["weights"])
    assert float(plain["flat"][0].abs().max()) > 0 and np.isfinite(plain["loss"]).all()


def test_bench_distributed_branch_through_rccl_one_rank(tmp_path):
    """bench.py's N > 1 code (NCCL barrier + max-over-ranks around the timed region, the strong leg, the training leg
    with and without the gradient all-reduce) at world size 1 through RCCL."""
    r = _run([os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--diffusion-steps", "20",
              "--batch", "64", "--no-cpu-baseline", "--no-alt-math", "--train-batch", "32", "--train-steps", "3",
              "--legs", "--infer-driver-rows", "300", "--train-driver-rows", "2000", "--train-driver-batch", "96"],
             29563, True, REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["strong"]["scaling"] == "strong" and out["strong"]["global_batch"] == 64 and out["strong"]["value"] > 0
    tr = out["train"]
    assert tr["value"] > 0 and 0.0 <= tr["allreduce_share"] < 1.0 and np.isfinite(tr["loss"])
    assert "RCCL" in tr["allreduce"]
    # the driver legs ran with the process group live: infer.py's final gather (all_gather on device tensors) and
    # train.py's own loop (seed broadcast, bucket all-reduce, barrier) through RCCL -- no error swallowed
    assert out["infer_driver"]["value"] > 0 and out["infer_driver"]["series"] == 300, out["infer_driver"]
    assert out["train_driver"]["value"] > 0 and np.isfinite(out["train_driver"]["loss"]), out["train_driver"]
