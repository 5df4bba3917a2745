"""Host logic of the drivers that needs no GPU: the loader-order helpers the resident data paths of infer.py / train.py
stand on (they must visit exactly the rows a pass over the DataLoader visits, and consume the CPU generator the same
way), the launch plan of infer.py, and bench.py's self-spawn launcher."""
import io
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from datafactory.dataloader import (AlternatingDataset, custom_collate_fn, epoch_index_batches, group_by_dataset,
                                    loader_provider, plan_epochs, resident_tables, walk_index_batches)
from datafactory.dataset import SyntheticT2SDataset

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    base = dict(dataset_name="ETTh1_24", batch_size=4, mix_train=False, synthetic=11)
    base.update(kw)
    return types.SimpleNamespace(**base)


@pytest.mark.parametrize("mix,bs", [(False, 4), (False, 1), (True, 7)])
def test_epoch_index_batches_is_one_pass_over_the_loader(mix, bs):
    """Same rows in the same order as `for data in loader` (reference dataloader.py:99,111: shuffle=True, drop_last=True),
    for three consecutive passes, and the global CPU generator ends in the same state (train.py's t / CFG-coin draws
    follow it)."""
    ds, loader = loader_provider(_args(mix_train=mix, batch_size=bs, dataset_name="ETTh1" if mix else "ETTh1_24"), "train")
    torch.manual_seed(77)
    real = []
    for _ in range(3):
        for data in loader:
            if mix:
                real.append(np.concatenate([g[1].numpy() for g in data if g[1].shape[1] == 24] or [np.zeros((0, 24))]))
            else:
                real.append(data[1].numpy())
    after_real = float(torch.rand(1))
    torch.manual_seed(77)
    tabs = resident_tables(ds)
    starts = [st for _, _, st in tabs]
    got = []
    for _ in range(3):
        batches = epoch_index_batches(loader)
        assert batches.shape == (len(loader), bs)
        for idx in batches:
            if mix:
                rows = [r for w, r in group_by_dataset(idx, starts) if w == 0]
                got.append(tabs[0][0][rows[0].numpy()] if rows else np.zeros((0, 24)))
            else:
                got.append(tabs[0][0][idx.numpy()])
    assert float(torch.rand(1)) == after_real, "the helper consumed the global generator differently from a real pass"
    assert len(real) == len(got) == 3 * len(loader)
    for a, b in zip(real, got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bs", [4, 1])
def test_walked_index_batches_equal_the_emulated_draws_and_a_real_pass(bs):
    """`--loader_batches` of infer.py: the order from a DataLoader WALK over the row numbers (public torch API only) is the
    order epoch_index_batches emulates and the order of a real pass, over consecutive passes, generator state included --
    the hard gate against a torch upgrade that changes how DataLoader / RandomSampler draw."""
    ds, loader = loader_provider(_args(batch_size=bs), "test")
    tab = resident_tables(ds)[0][0]
    runs = {}
    for name, fn in (("emulated", epoch_index_batches), ("walked", walk_index_batches)):
        torch.manual_seed(123)
        runs[name] = ([fn(loader) for _ in range(3)], float(torch.rand(1)))
    torch.manual_seed(123)
    real = [np.stack([d[1].numpy() for d in loader]) for _ in range(3)]
    after_real = float(torch.rand(1))
    assert runs["emulated"][1] == runs["walked"][1] == after_real
    for a, b, r in zip(runs["emulated"][0], runs["walked"][0], real):
        assert a.dtype == b.dtype == torch.int64 and torch.equal(a, b)
        assert np.array_equal(tab[a.numpy()], r)


def test_emulated_draws_refuse_a_loader_with_its_own_generator():
    """RandomSampler with an explicit generator draws differently (a trailing randperm from it): not emulated, refused --
    walk_index_batches serves that shape."""
    ds = SyntheticT2SDataset(11, 24)
    g = torch.Generator().manual_seed(3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=True, drop_last=True, generator=g)
    with pytest.raises(ValueError, match="explicit generator"):
        epoch_index_batches(loader)
    g.manual_seed(3)
    walked = walk_index_batches(loader)
    g.manual_seed(3)
    real = np.stack([d[1].numpy() for d in loader])
    assert np.array_equal(ds.samples[walked.numpy()], real)


@pytest.mark.parametrize("mix,bs,n_rows,epochs,min_steps", [(True, 7, 11, 9, 10), (False, 4, 11, 7, 5), (True, 5, 3, 12, 4)])
def test_epoch_plans_replay_the_loader_passes_and_coins_across_plan_boundaries(mix, bs, n_rows, epochs, min_steps):
    """train.py's resident path uploads index batches as multi-epoch PLANS (datafactory.plan_epochs).  Walking plan after
    plan must visit, step by step, the rows the reference-style loop visits -- `for epoch: for data in loader: for group in
    data: coin = torch.rand(1) < 0.3` (train.py:52-95 / 101-131) -- with the same coins, across epoch AND plan boundaries
    (plans of several epochs, a plan that ends exactly at an epoch, more plans than one), and leave the CPU generator in the
    same state."""
    args = _args(mix_train=mix, batch_size=bs, dataset_name="ETTh1" if mix else "ETTh1_24", synthetic=n_rows)
    ds, loader = loader_provider(args, "train")
    tabs = resident_tables(ds)
    starts = [st for _, _, st in tabs]
    torch.manual_seed(5)
    want = []                                                     # (epoch, rows of the group as series, coin)
    for e in range(epochs):
        for data in loader:
            for g in (data if mix else [data]):
                want.append((e, g[1].numpy(), bool(torch.rand(1) < 0.3)))
    after_real = float(torch.rand(1))
    torch.manual_seed(5)
    got, e, n_plans = [], 0, 0
    while e < epochs:
        plans, flat = plan_epochs(loader, starts, mix, e, epochs, min_steps)
        assert plans and [p[0] for p in plans] == list(range(e, e + len(plans)))
        assert sum(len(g) for p in plans[:-1] for g in p[2]) < min_steps          # stops as soon as the plan is long enough
        off = 0
        for ep, rows, groups, coins in plans:
            for b in range(rows.shape[0]):
                o = off + b * rows.shape[1]
                for (w, c), coin in zip(groups[b], coins[b]):
                    got.append((ep, tabs[w][0][flat[o:o + c].numpy()], coin))
                    o += c
                assert o == off + (b + 1) * rows.shape[1]
            off += rows.numel()
        assert off == flat.numel()
        e += len(plans)
        n_plans += 1
    assert float(torch.rand(1)) == after_real
    assert n_plans > 1 and len(got) == len(want) > 0
    for (ea, ra, ca), (eb, rb, cb) in zip(want, got):
        assert ea == eb and ca == cb and np.array_equal(ra, rb)


def test_resident_tables_on_the_csv_dataset_equal_a_loader_pass(tmp_path, monkeypatch):
    """The same order / table helpers on the real CSV path (T2SDataset, reference datafactory/dataset.py:10-104): rows and
    embeddings gathered from resident_tables by epoch_index_batches equal what a pass over loader_provider's DataLoader
    collates, float64 -> `.float()` included (the conversion infer.py:70-71 / train.py:104-105 apply per batch)."""
    import shutil
    root = tmp_path / "Data" / "our"
    os.makedirs(root)
    shutil.copy(os.path.join(REPO, "tests", "golden", "dataset_csv", "embedding_cleaned_ETTh1_24.csv"),
                root / "embedding_cleaned_ETTh1_24.csv")
    monkeypatch.chdir(tmp_path)
    ds, loader = loader_provider(_args(dataset_name="ETTh1_24", batch_size=7, synthetic=0), "train")
    (series, emb, start), = resident_tables(ds)
    assert start == 0 and series.shape == (60, 24) and emb.shape == (60, 128)
    torch.manual_seed(3)
    real = [(x.float(), e.float()) for _, x, e in loader]
    torch.manual_seed(3)
    batches = epoch_index_batches(loader)
    assert len(real) == batches.shape[0] == 60 // 7
    xs, es = torch.as_tensor(series).float(), torch.as_tensor(emb).float()
    for (x, e), idx in zip(real, batches):
        assert torch.equal(x, xs[idx]) and torch.equal(e, es[idx])


def test_group_by_dataset_is_the_collate_grouping():
    """custom_collate_fn (dataloader.py:115-133) groups a mixed batch by source dataset, keeping batch order inside a
    group and skipping empty groups: the index-space version must give the same rows and embeddings."""
    parts = [SyntheticT2SDataset(5, L, seed=L) for L in (24, 48, 96)]
    ds = AlternatingDataset(*parts)
    tabs = resident_tables(ds)
    starts = [st for _, _, st in tabs]
    assert starts == [0, 5, 10]
    for idx in (torch.tensor([14, 0, 7, 3, 9, 1]), torch.tensor([12, 11]), torch.tensor([4]), torch.tensor([5, 10, 0])):
        want = custom_collate_fn([ds[int(i)] for i in idx])
        got = group_by_dataset(idx, starts)
        assert len(want) == len(got)
        for (texts, xs, embs), (w, rows) in zip(want, got):
            assert np.array_equal(xs.numpy(), tabs[w][0][rows.numpy()])
            assert np.array_equal(embs.numpy(), tabs[w][1][rows.numpy()])
            assert texts == [parts[w].text[int(r)] for r in rows]


def test_infer_launch_plan_covers_rows_once_in_order():
    pytest.importorskip("torch")
    import infer
    for n_rows, bs, lb, world in ((2048, 2, 256, 1), (600, 2, 256, 1), (37, 1, 256, 8), (40, 4, 0, 2), (4096, 2, 256, 8)):
        plan = infer.launch_plan(n_rows, bs, lb, world)
        assert plan[0][0] == 0 and plan[-1][1] == n_rows
        assert all(b == c for (_, b), (c, _) in zip(plan, plan[1:]))
        per = bs if lb == 0 else lb * world
        assert all(s1 - s0 == per for s0, s1 in plan[:-1]) and 0 < plan[-1][1] - plan[-1][0] <= per


def test_bench_self_spawn_builds_a_fresh_rank_launcher(monkeypatch, capsys):
    """`python bench.py --gpus N` with no torchrun environment starts `python -m torch.distributed.run --nproc-per-node N
    bench.py <same flags>` as a CHILD (never exec), relays rank 0's one JSON line and returns the child's exit code."""
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO("rank chatter\n" + json.dumps({"metric": "m", "n_gpus": 4}) + "\nlate chatter\n")

        def wait(self):
            return seen.get("rc", 0)

    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setenv("RANK", "3")               # stale torchrun variables must not leak into the children
    rc = bench.spawn_ranks(4, ["--gpus", "4", "--steps", "2"])
    out = capsys.readouterr().out.splitlines()
    assert rc == 0 and out[-1] == json.dumps({"metric": "m", "n_gpus": 4}) and "rank chatter" in out
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "2"]
    assert "RANK" not in seen["env"] and "WORLD_SIZE" not in seen["env"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen["rc"] = 7                                 # a failing child: its code comes back, no result line is printed
    assert bench.spawn_ranks(4, ["--gpus", "4"]) == 7
    assert not [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]


def test_shipped_default_math_is_bf16x3_and_the_suite_pins_f32(monkeypatch):
    """Round 5 made the fp32-accurate bf16x3 arithmetic the default of the Sampler and of infer.py (profiles/r05_accuracy.md);
    the bench headline and the class API stay on the exact f32 MFMA.  The suite itself runs with T2S_DEFAULT_MATH=f32."""
    import infer
    from t2ms_amd import sampler
    assert sampler.DEFAULT_MATH == "bf16x3"
    assert os.environ.get("T2S_DEFAULT_MATH") == "f32" and sampler.default_math() == "f32"
    monkeypatch.delenv("T2S_DEFAULT_MATH")
    assert sampler.default_math() == "bf16x3"
    assert infer.build_parser().parse_args([]).math is None                  # resolved by default_math() at run time
    assert infer.build_parser().parse_args(["--math", "f32"]).math == "f32"
    monkeypatch.setenv("T2S_DEFAULT_MATH", "tf32")
    with pytest.raises(ValueError):
        sampler.default_math()
