"""The N>1 code paths of train.py / infer.py / bench.py through the REAL HIP kernels, rehearsed as two ranks on the
one GPU of the test box: fresh child processes under torch.distributed.run, both on cuda:0 (T2S_SHARE_GPU=1),
collectives over gloo (T2S_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  SURVEY.md 8(e).

What must hold (and is checked against a single-process run of the same work):
  * training: the all-reduced flat gradient bucket equals the single-process gradient of the whole batch, the ranks
    end with bit-identical weights, ragged (3 + 2) and empty (1 + 0) shards included;
  * sampling: the sharded .npy files equal the single-process files bitwise (Philox keyed by the global row);
  * bench.py --gpus 2 prints ONE JSON line with n_gpus = 2.
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


def _env(port, two_ranks):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    if two_ranks:
        env.update(T2S_DIST_BACKEND="gloo", T2S_SHARE_GPU="1")
    return env


def _launch(script_and_args, port, two_ranks, cwd, timeout=420, nproc=2):
    if two_ranks:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_and_args
    else:
        cmd = [sys.executable] + script_and_args
    r = subprocess.run(cmd, env=_env(port, two_ranks), cwd=cwd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r


TRAIN_WORKER = textwrap.dedent("""
    import os, sys, types
    import torch
    import train as T                                   # the drop-in driver at the repo root
    from t2ms_amd import dist as tdist, synth
    from t2ms_amd.train import T2SAdamW, N_GRAD, _trainable
    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    from model.backbone.DDPM import DDPM

    out_dir = sys.argv[1]
    rank, _, world = tdist.env_world()
    torch.cuda.set_device(tdist.local_device_index())
    dev = torch.device("cuda", tdist.local_device_index())
    dist = tdist.init("nccl", dev)
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
    # steps 0, 1: 5 rows (ragged 3 + 2 under two ranks); step 2: ONE row (rank 1's shard is empty); step 3: 4 rows
    for step_no, B in enumerate((5, 5, 1, 4)):
        x = synth.make_series(10 + step_no, B, 96)
        emb = synth.make_text_embeddings(10 + step_no, B)
        loss = T.train_step(m, ddpm, opt, dist, args, x, emb, dev, rank, world, None, None, step_no)
        rec["flat"].append(m.__dict__["_t2s_flat_grad"][:N_GRAD].detach().cpu().clone())
        rec["loss"].append(float(loss))
    rec["weights"] = torch.cat([p.detach().reshape(-1).cpu() for p in _trainable(m)])
    torch.cuda.synchronize()
    torch.save(rec, os.path.join(out_dir, f"train_w{world}_r{rank}.pt"))
    tdist.barrier(dist, dev)
    print("TRAIN WORKER OK", rank, world)
""")


def test_two_rank_training_equals_single_process(tmp_path):
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    _launch([str(script), str(tmp_path)], 29551, False, REPO)
    _launch([str(script), str(tmp_path)], 29552, True, REPO)
    one = torch.load(tmp_path / "train_w1_r0.pt")
    r0 = torch.load(tmp_path / "train_w2_r0.pt")
    r1 = torch.load(tmp_path / "train_w2_r1.pt")
    for step in range(4):
        g1, g2 = one["flat"][step], r0["flat"][step]
        assert torch.equal(r0["flat"][step], r1["flat"][step]), f"step {step}: ranks hold different reduced buckets"
        scale = float(g1.abs().max())
        assert scale > 0
        err = float((g1 - g2).abs().max())
        # step 0 starts from identical weights: the all-reduced bucket IS the whole-batch gradient (fp32 summation order
        # differs: per-shard sums then a weighted add).  Later steps compare after AdamW updates that agree to ~1e-7.
        assert err <= (1e-6 if step == 0 else 2e-5) * scale, (step, err, scale)
        assert abs(one["loss"][step] - r0["loss"][step]) <= 1e-6 * max(1.0, abs(one["loss"][step])), (step, one["loss"], r0["loss"])
        assert r0["loss"][step] == r1["loss"][step]
    assert torch.equal(r0["weights"], r1["weights"]), "ranks diverged: weights are not bit-identical"
    assert float((one["weights"] - r0["weights"]).abs().max()) <= 5e-4   # 4 AdamW steps of lr 1e-4 each


def test_two_rank_sampling_files_equal_single_process_bitwise(tmp_path):
    argv = ["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "4",
            "--cfg_scale", "9", "--batch_size", "5", "--synthetic", "16", "--random_init", "--seed", "11"]
    sub = os.path.join("generation", "ddpm_DiT_ETTh1_24_9.0_4")
    _launch([os.path.join(REPO, "infer.py")] + argv + ["--save_path", str(tmp_path / "one")], 29553, False, str(tmp_path))
    _launch([os.path.join(REPO, "infer.py")] + argv + ["--save_path", str(tmp_path / "two")], 29554, True, str(tmp_path))
    for f in ("x_1.npy", "x_t.npy", "x_t_latent_dec_array.npy", "x_t_latent_enc_array.npy"):
        a = np.load(tmp_path / "one" / sub / f)
        b = np.load(tmp_path / "two" / sub / f)
        assert a.shape == b.shape and a.shape[0] == 15 and np.isfinite(a).all()
        assert np.array_equal(a, b), f"{f}: sharded output differs from the single-process output"


def test_two_rank_sampling_with_empty_shards_and_coalesced_launches(tmp_path):
    """--batch_size 1 (and the reference's default 2) on two ranks.  With one launch per loader batch (`--launch_batch 0`)
    every launch has ONE row: rank 1 never samples and must still reach the final gather (round 3 raised
    `t2s_vae_encode: B=0` there while rank 0 blocked in all_gather).  With coalesced launches both ranks sample.  All
    variants must write the single-process files bit for bit (reference infer.py:66,129: drop_last loader, default batch 2)."""
    base = ["--dataset_name", "ETTh1_24", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "3", "--cfg_scale", "9",
            "--synthetic", "7", "--random_init", "--seed", "4"]
    sub = os.path.join("generation", "ddpm_DiT_ETTh1_24_9.0_3")
    files = ("x_1.npy", "x_t.npy", "x_t_latent_dec_array.npy", "x_t_latent_enc_array.npy")
    port = 29580
    for bs in ("1", "2"):
        _launch([os.path.join(REPO, "infer.py")] + base + ["--batch_size", bs, "--save_path", str(tmp_path / f"one{bs}")],
                port, False, str(tmp_path))
        want = [np.load(tmp_path / f"one{bs}" / sub / f) for f in files]
        assert want[0].shape[0] == (7 // int(bs)) * int(bs)
        for lb in (("0", "256") if bs == "1" else ("0",)):
            port += 1
            _launch([os.path.join(REPO, "infer.py")] + base + ["--batch_size", bs, "--launch_batch", lb, "--save_path",
                                                               str(tmp_path / f"two{bs}_{lb}")], port, True, str(tmp_path))
            for f, a in zip(files, want):
                assert np.array_equal(a, np.load(tmp_path / f"two{bs}_{lb}" / sub / f)), (bs, lb, f)
        port += 1


def test_four_ranks_with_fewer_rows_than_ranks(tmp_path):
    """Four ranks on the one test GPU (the most this box allows beside the test process), a launch of THREE rows: ranks 0-2
    sample one row each, rank 3 none -- in infer.py (final gather with an empty-handed rank, files bitwise equal to the
    single-process run) and in train.py's own loop (length groups of 1-3 rows: up to three ranks contribute zero buckets to
    the all-reduce and still step their optimizer; the run finishes and writes the checkpoint)."""
    base = ["--dataset_name", "ETTh1_24", "--backbone", "flowmatching", "--denoiser", "DiT", "--total_step", "3", "--cfg_scale", "7",
            "--synthetic", "3", "--random_init", "--seed", "8", "--batch_size", "3"]
    sub = os.path.join("generation", "flowmatching_DiT_ETTh1_24_7.0_3")
    _launch([os.path.join(REPO, "infer.py")] + base + ["--save_path", str(tmp_path / "one")], 29570, False, str(tmp_path))
    _launch([os.path.join(REPO, "infer.py")] + base + ["--save_path", str(tmp_path / "four")], 29571, True, str(tmp_path), nproc=4)
    for f in ("x_1.npy", "x_t.npy", "x_t_latent_dec_array.npy", "x_t_latent_enc_array.npy"):
        a, b = np.load(tmp_path / "one" / sub / f), np.load(tmp_path / "four" / sub / f)
        assert a.shape[0] == 3 and np.array_equal(a, b), f
    argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100", "--batch_size", "4",
            "--epochs", "2", "--save_path", str(tmp_path / "res"), "--synthetic", "4", "--random_init", "--checkpoint_path", "",
            "--bf16"]
    _launch([os.path.join(REPO, "train.py")] + argv, 29572, True, str(tmp_path), timeout=600, nproc=4)
    ck = torch.load(tmp_path / "res" / "checkpoints" / "ddpm_DiT_ETTh1" / "model_1.pth", map_location="cpu")
    assert ck["epoch"] == 1 and len(ck["loss_list"]) >= 6 and np.isfinite(ck["loss_list"]).all()


def test_two_rank_default_seed_is_rank0s(tmp_path):
    """No --seed: every rank must end up with rank 0's time-based seed (loader order + Philox key)."""
    script = tmp_path / "seed_worker.py"
    script.write_text(textwrap.dedent("""
        import json, sys, time
        import infer as I
        from t2ms_amd import dist as tdist
        rank, _, world = tdist.env_world()
        time.sleep(1.3 * rank)                      # the ranks read the clock in different seconds
        seen = {}
        real = I.loader_provider
        def spy(a, period):
            seen["seed"] = a.seed                   # what seeds the loader shuffle and keys the Philox noise
            return real(a, period)
        I.loader_provider = spy
        I.main(["--synthetic", "8", "--random_init", "--batch_size", "4", "--total_step", "2", "--backbone", "ddpm",
                "--dataset_name", "ETTh1_24", "--save_path", sys.argv[1]])
        open(f"{sys.argv[1]}.seed{rank}", "w").write(json.dumps(seen))
    """))
    _launch([str(script), str(tmp_path / "o")], 29555, True, REPO)
    a = json.load(open(f"{tmp_path / 'o'}.seed0"))
    b = json.load(open(f"{tmp_path / 'o'}.seed1"))
    assert a["seed"] == b["seed"], (a, b)


def test_two_rank_train_cli_with_groups_smaller_than_the_world(tmp_path):
    """train.py itself under two ranks, mix-train (three length groups per batch) with batches so small that groups of ONE
    row occur: the rank without rows must still join the all-reduce (round 1 returned early there: an RCCL hang).  The
    run has to finish and rank 0 has to write the reference checkpoint dict."""
    argv = ["--dataset_name", "ETTh1", "--backbone", "ddpm", "--denoiser", "DiT", "--total_step", "100", "--batch_size", "4",
            "--epochs", "2", "--save_path", str(tmp_path / "res"), "--synthetic", "5", "--random_init", "--checkpoint_path", ""]
    r = _launch([os.path.join(REPO, "train.py")] + argv, 29557, True, str(tmp_path), timeout=600)
    ck = torch.load(tmp_path / "res" / "checkpoints" / "ddpm_DiT_ETTh1" / "model_1.pth", map_location="cpu")
    assert set(ck) == {"model", "optimizer", "epoch", "loss_list"} and ck["epoch"] == 1
    assert len(ck["loss_list"]) >= 6 and np.isfinite(ck["loss_list"]).all(), r.stdout[-1500:]
    assert len(ck["optimizer"]["param_groups"][0]["params"]) == 67 and len(ck["optimizer"]["state"]) == 48


def test_plain_bench_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver types it -- no torchrun, no WORLD_SIZE: the process must start the
    two ranks itself (fresh children, before any GPU call of its own), relay ONE JSON line with n_gpus = 2 and exit 0."""
    env = _env(0, True)
    env.pop("MASTER_ADDR"), env.pop("MASTER_PORT")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--diffusion-steps", "10", "--batch", "64", "--no-cpu-baseline", "--no-train", "--no-legs"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["value"] > 0 and out["strong"]["value"] > 0


def test_bench_two_ranks_prints_one_json_line(tmp_path):
    r = _launch([os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--diffusion-steps", "20",
                 "--batch", "64", "--no-cpu-baseline", "--train-batch", "32", "--train-steps", "3",
                 "--legs", "--infer-driver-rows", "300", "--train-driver-rows", "2000", "--train-driver-batch", "96"], 29556, True, REPO)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["value"] > 0
    assert out["scaling"] == "weak" and out["steps"] == 1
    # strong scaling next to it: the same 64 series split over the two ranks (32 each)
    st = out["strong"]
    assert st["scaling"] == "strong" and st["global_batch"] == 64 and st["per_gpu_batch"] == 32 and st["value"] > 0
    tr = out["train"]                 # the training leg ran on both ranks with the gradient all-reduce in it
    assert tr["global_batch"] == 64 and tr["value"] > 0 and 0.0 <= tr["allreduce_share"] < 1.0 and np.isfinite(tr["loss"])
    # the real drivers under two ranks: infer.py (launches sharded over the ranks, ONE final gather) and train.py's loop
    assert out["infer_driver"]["n_gpus"] == 2 and out["infer_driver"]["series"] == 300 and out["infer_driver"]["value"] > 0
    assert out["train_driver"]["n_gpus"] == 2 and out["train_driver"]["value"] > 0, out["train_driver"]
