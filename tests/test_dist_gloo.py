"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding / barrier / max-time /
gather glue bench.py and infer.py use (the GPU data path itself has no collective)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from t2ms_amd import dist as tdist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_rows_partition():
    for total in (1, 7, 256, 257, 1024):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_rows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import numpy as np, torch
    from t2ms_amd import dist as tdist, synth
    from oracle import t2s_oracle as O
    rank, local_rank, world = tdist.env_world()
    d = tdist.init("gloo")
    assert d is not None and d.get_world_size() == world == 2
    total = 7
    lo, hi = tdist.shard_rows(total, rank, world)
    # every rank draws ITS rows of the global Philox stream (oracle restatement of the device RNG)
    local = torch.from_numpy(O.device_normal(2025, 3, lo, hi - lo))
    text = synth.make_text_embeddings(2025, hi - lo, row0=lo)
    tdist.barrier(d)
    t = tdist.max_over_ranks(d, 1.0 + rank)
    assert t == 2.0, t
    full = tdist.gather_rows(d, local, total, rank, world)
    ftext = tdist.gather_rows(d, text, total, rank, world)
    # one seed for the whole job: rank 0's value wins (infer.py: a per-rank time-based default mis-pairs series)
    seed = tdist.broadcast_int(d, 1000 + 17 * rank)
    assert seed == 1000, seed
    # the training all-reduce: ONE flat bucket, weighted by shard rows (t2ms_amd.train.allreduce_gradients)
    from t2ms_amd.train import allreduce_gradients, grad_bucket, _trainable, N_GRAD
    from model.denoiser.transformer import Transformer
    m = Transformer()
    ts = _trainable(m)
    assert len(ts) == 48 and sum(t.numel() for t in ts) == N_GRAD == 925592
    # ragged shards 3 + 1 rows: rank r's mean-loss gradient is the constant r + 1, its loss 10 (r + 1)
    n_local = 3 if rank == 0 else 1
    for p in ts:
        p.grad = torch.full_like(p, float(rank + 1))       # NOT aliasing the bucket: must be moved into it
    flat, loss = allreduce_gradients(m, d, n_local=n_local, n_global=4, loss=torch.tensor(10.0 * (rank + 1)))
    want = (3 * 1.0 + 1 * 2.0) / 4
    assert torch.all(flat[:N_GRAD] == want), flat[:4]
    assert abs(float(loss) - 10 * want) < 1e-6
    b = grad_bucket(m, torch.device("cpu"))
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ts, b["views"]))      # p.grad aliases the bucket
    assert all(torch.all(p.grad == want) for p in ts)
    # a rank with an EMPTY shard still joins the collective with a zero bucket (no hang, same result on both)
    for p in ts:
        p.grad = None if rank == 1 else torch.full_like(p, 4.0)
    flat, loss = allreduce_gradients(m, d, n_local=1 if rank == 0 else 0, n_global=1,
                                     loss=torch.tensor(2.0 if rank == 0 else 0.0))
    assert torch.all(flat[:N_GRAD] == 4.0) and abs(float(loss) - 2.0) < 1e-6
    assert all(p.grad is not None and torch.all(p.grad == 4.0) for p in ts)
    # default weights (no row counts): the plain average
    for p in ts:
        p.grad = torch.full_like(p, float(rank + 1))
    flat, _ = allreduce_gradients(m, d)
    assert torch.all(flat[:N_GRAD] == 1.5), flat[:4]
    # infer.py's ONE final gather: ragged per-rank row counts, a rank may hold nothing at all
    for counts in ([3, 2], [4, 0], [0, 1]):
        mine = torch.full((counts[rank], 5), float(rank + 1))
        parts = tdist.gather_ragged(d, mine, counts, rank)
        if rank == 0:
            assert [tuple(p.shape) for p in parts] == [(c, 5) for c in counts]
            assert all(torch.all(p == r + 1) for r, p in enumerate(parts))
        else:
            assert parts is None
    if rank == 0:
        ref = torch.from_numpy(O.device_normal(2025, 3, 0, total))
        assert torch.equal(full, ref), "sharded draws differ from the single-process stream"
        assert torch.equal(ftext, synth.make_text_embeddings(2025, total))
        print("OK", tuple(full.shape))
    else:
        assert full is None
    tdist.barrier(d)
    d.destroy_process_group()
""")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK (7, 1920)" in r.stdout


def test_optimizer_state_interchanges_with_torch_adamw():
    """train.py:37 builds AdamW(model.parameters()): 67 tensors, pos_embed first, the frozen encoder last.  A state
    dict written by torch.optim.AdamW over that list must load into T2SAdamW built the way train.py builds it, and
    the other way round (the checkpoint's `optimizer` entry, train.py:44,94)."""
    import types

    import torch

    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    from t2ms_amd.train import T2SAdamW, _trainable

    def make():
        m = Transformer()
        v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                        embedding_dim=64))
        m.encoder = v.encoder
        for n, p in m.named_parameters():
            if "encoder" in n:
                p.requires_grad = False
        return m

    ref_model = make()
    names = [n for n, _ in ref_model.named_parameters()]
    assert len(names) == 67 and names[0] == "pos_embed" and names[-1].startswith("encoder.")
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=1e-4, weight_decay=0.0)
    for p in _trainable(ref_model):                     # the 48 tensors that receive gradients
        p.grad = torch.randn_like(p)
    ref_opt.step()
    sd = ref_opt.state_dict()
    assert len(sd["state"]) == 48 and len(sd["param_groups"][0]["params"]) == 67
    ours = T2SAdamW(make().parameters(), lr=1e-4, weight_decay=0.0)
    ours.load_state_dict(sd)                            # raised ValueError when only the trainable subset was passed
    back = ours.state_dict()
    assert back["param_groups"][0]["params"] == sd["param_groups"][0]["params"]
    assert sorted(back["state"]) == sorted(sd["state"])
    for k in sd["state"]:
        assert set(back["state"][k]) == {"step", "exp_avg", "exp_avg_sq"}
        assert torch.equal(back["state"][k]["exp_avg"], sd["state"][k]["exp_avg"])
    torch.optim.AdamW(make().parameters(), lr=1e-4, weight_decay=0.0).load_state_dict(back)
