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
    # the training all-reduce: ONE flat bucket, averaged (t2ms_amd.train.allreduce_gradients)
    import types as _types
    from t2ms_amd.train import allreduce_gradients, N_GRAD
    fake = _types.SimpleNamespace()
    fake.__dict__["_t2s_flat_grad"] = torch.full((N_GRAD,), float(rank + 1))
    avg = allreduce_gradients(fake, d)
    assert avg.numel() == 925592 and torch.all(avg == 1.5), avg[:4]
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
