"""CPU checks against the round-2 fixtures that tests/golden/gen_golden_r2.py made by RUNNING THE REFERENCE:
seeded construction, the dataset path (8f row 3), the evaluation metrics (8f row 4), the TS2Vec
encoder, the 1000-step DDPM chain at the headline schedule (north_star: fp32 1e-4 on fixed seeds)."""
import os
import shutil
import types

import numpy as np
import pytest
import torch

from oracle import t2s_oracle as O
from t2ms_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


# ------------------------------------------------------------------------------------------- a15: seeded init
@pytest.mark.parametrize("seed", [0, 2025])
def test_seeded_construction_equals_reference(seed):
    """transformer.py:128-154,194-204 under torch.manual_seed(k): same parameter ORDER (optimizer state indices,
    train.py:37) and the same VALUES -- the constructors must consume the generator in the reference's order
    (conv, patch_emb, ln, linear_emb_to_patch, 4 blocks [attn.qkv, attn.proj, mlp.fc1, mlp.fc2, adaLN], unpatch),
    then initialize_weights re-draws every Linear with xavier_uniform in module order and zeroes the adaLN layer."""
    from model.denoiser.transformer import Transformer
    g = _load("init_seeded")
    torch.manual_seed(seed)
    m = Transformer()
    names = [n for n, _ in m.named_parameters()]
    assert names == [str(n) for n in g[f"names_{seed}"]]
    for n, p in m.named_parameters():
        key = n.replace(".", "__")
        v = p.detach().double().flatten()
        stat = np.asarray([float(v.sum()), float((v * v).sum()), float(v.abs().max())])
        assert np.array_equal(p.detach().flatten()[::97].numpy(), g[f"samp_{seed}_{key}"]), n
        np.testing.assert_allclose(stat, g[f"stat_{seed}_{key}"], rtol=1e-12, atol=0, err_msg=n)
    assert np.array_equal(torch.rand(4).numpy(), g[f"rng_after_{seed}"])     # same generator consumption
    assert float(m.layers[0].attn.qkv.weight.abs().max()) > 0 and float(m.layers[3].adaLN_modulation[-1].weight.abs().max()) == 0


# ------------------------------------------------------------------------------------------- f3: dataset path
@pytest.mark.parametrize("family", ["ETTh1", "Climate"])
def test_dataset_first_load_and_cache_equal_reference(tmp_path, family):
    """datafactory.T2SDataset (first load: CSV parse; second load: the .t2scache.npz it wrote) against what the
    reference's T2SDataset (datafactory/dataset.py:10-104) made of the same CSV: MinMax-scaled series to 1 ulp,
    embeddings bit for bit in both TextEmbedding formats, texts, the seed-123 permutation split, item layout."""
    from datafactory.dataset import T2SDataset, split_indices
    g = _load("dataset")
    name = f"embedding_cleaned_{family}_24"
    shutil.copy(os.path.join(GOLD, "dataset_csv", name + ".csv"), tmp_path / (name + ".csv"))
    np.random.seed(5)
    before = np.random.get_state()[1][:8].copy()
    for attempt in ("csv", "cache"):
        for period in ("train", "test"):
            ds = T2SDataset(name=name, data_root=str(tmp_path), period=period, proportion=0.9)
            key = f"{family}_{period}"
            ref = g[f"samples_{key}"]
            assert ds.samples.shape == ref.shape and ds.samples.dtype == np.float64
            np.testing.assert_array_max_ulp(ds.samples, ref, maxulp=1)
            assert np.array_equal(np.asarray(ds.embedding, dtype=np.float64), g[f"embedding_{key}"]), (attempt, key)
            assert list(ds.text) == [str(t) for t in g[f"text_{key}"]]
            assert [len(ds), ds.len, ds.var_num] == list(g[f"meta_{key}"])
            t1, x1, e1 = ds[1]
            assert t1 == str(g[f"item1_t_{key}"])
            np.testing.assert_array_max_ulp(np.asarray(x1), g[f"item1_x_{key}"], maxulp=1)
            assert np.array_equal(np.asarray(e1), g[f"item1_e_{key}"])
        assert os.path.exists(tmp_path / (name + ".t2scache.npz"))
    assert np.array_equal(np.random.get_state()[1][:8], before)       # the split leaves the global numpy RNG alone
    # constant OT column -> 0 everywhere (sklearn MinMaxScaler's zero-range rule)
    assert float(np.abs(g[f"samples_{family}_train"][:, 5]).max()) == 0.0 and float(np.abs(ds.samples[:, 5]).max()) == 0.0
    # default 99 % split of 60 rows = ceil(59.4) = 60 train rows and an EMPTY test split, as in the reference
    tr, te = split_indices(60)
    assert [len(tr), len(te)] == list(g["default_split_lens"]) == [60, 0]


def test_loader_provider_contract(tmp_path, monkeypatch):
    """loader_provider (datafactory/dataloader.py:79-113): split-train loader over one CSV, shuffle + drop_last for both
    periods, item = (text, x (L,), embedding (128,))."""
    from datafactory import dataloader as DL
    root = tmp_path / "Data" / "our"
    os.makedirs(root)
    shutil.copy(os.path.join(GOLD, "dataset_csv", "embedding_cleaned_ETTh1_24.csv"), root / "embedding_cleaned_ETTh1_24.csv")
    monkeypatch.chdir(tmp_path)
    args = types.SimpleNamespace(dataset_name="ETTh1_24", batch_size=7, mix_train=False, synthetic=0)
    ds, loader = DL.loader_provider(args, "train")
    assert len(ds) == 60 and len(loader) == 60 // 7
    text, x, emb = next(iter(loader))
    assert len(text) == 7 and tuple(x.shape) == (7, 24) and tuple(emb.shape) == (7, 128) and x.dtype == torch.float64


# ------------------------------------------------------------------------------------------- f4: metrics
def test_metric_restatements_equal_reference():
    """oracle.eval_* against evaluation.py's calculate_mse / _wape / _mrr / _crps / _ed / calculate_fid and
    Evaluate_Datasets.cosine_similarity run by the generator (all-zero row, negated runs, an exact copy included)."""
    g = _load("metrics")
    ori, gen, runs = g["ori"], g["gen"], g["runs"]
    np.testing.assert_allclose(O.eval_mse(ori, gen), float(g["mse"]), rtol=1e-6)       # the reference sums in fp32
    np.testing.assert_allclose(O.eval_wape(ori, gen), float(g["wape"]), rtol=1e-6)
    assert abs(O.eval_mrr(ori, runs) - float(g["mrr"])) < 1e-12
    np.testing.assert_allclose(O.eval_crps(ori, runs), float(g["crps"]), rtol=1e-9)
    np.testing.assert_allclose(O.eval_ed(ori, gen), float(g["ed"]), rtol=1e-6)
    sims = np.asarray([[O.eval_cosine(ori[i], runs[i, :, :, k]) for k in range(runs.shape[3])] for i in range(ori.shape[0])])
    np.testing.assert_allclose(sims, g["sims"], rtol=1e-6, atol=1e-7)
    assert np.all(sims[4] == 0) and np.all(sims[2, :6] < 0) and abs(sims[6, 7] - 1) < 1e-6
    np.testing.assert_allclose([O.eval_cosine([1, 0, 0], [1, 1, 0]), O.eval_cosine([0, 0], [1, 1]),
                                O.eval_cosine(ori[0], gen[0])], g["cos_pairs"], rtol=1e-6)
    np.testing.assert_allclose(O.eval_fid(g["fid_act1"], g["fid_act2"]), float(g["fid"]), rtol=1e-9)


def test_dtw_restatement_known_answers():
    """dtaidistance is absent (UNPINNED): the restated DTW on cases with known answers -- identical series 0; a pure
    time shift of a step costs nothing; a constant offset d over L points costs d * sqrt(L)."""
    a = np.zeros((1, 8, 1))
    a[0, 3:, 0] = 1.0
    b = np.zeros((1, 8, 1))
    b[0, 5:, 0] = 1.0
    assert O.eval_dtw(a, a) == 0.0 and O.eval_dtw(a, b) == 0.0
    c = np.linspace(0, 1, 6).reshape(1, 6, 1) * 0 + 0.25
    np.testing.assert_allclose(O.eval_dtw(c, c + 0.5), 0.5 * np.sqrt(6), rtol=1e-12)


def test_ts2vec_encoder_restatement_equals_reference():
    """oracle.ts2vec_encode against TSEncoder.forward (evaluate/ts2vec.py:366-399) + the full-series max pooling, on
    the seeded weights of synth.make_ts2vec_state_dict (initialize_ts2vec's sizes: 1 -> 64 x 10 blocks -> 100)."""
    g = _load("ts2vec")
    sd = synth.make_ts2vec_state_dict(2025)
    with torch.no_grad():
        rep, full = O.ts2vec_encode(sd, torch.from_numpy(g["x"]))
    assert tuple(rep.shape) == (5, 96, 100)
    scale = float(np.abs(g["rep"]).max())
    assert scale > 0.1
    assert float(np.abs(rep[:, ::6].numpy() - g["rep"]).max()) <= 1e-5 * scale
    assert float(np.abs(full.numpy() - g["full_series"]).max()) <= 1e-5 * scale


# ------------------------------------------------------------------------------------------- attention branches
def test_oracle_attention_branches_agree():
    """timm's fused (SDPA) and explicit branches, as the oracle restates them, agree to fp32 rounding."""
    sd = synth.make_dit_state_dict(2025)
    x, text, t = synth.make_latents(3, 3), synth.make_text_embeddings(3, 3), torch.tensor([0, 500, 999])
    with torch.no_grad():
        a = O.dit_forward(sd, x, t, text)
        O.set_attention_impl("sdpa")
        try:
            b = O.dit_forward(sd, x, t, text)
        finally:
            O.set_attention_impl("explicit")
    assert float((a - b).abs().max()) < 2e-5


# ------------------------------------------------------------------------------------------- 1000-step chain
from _chain1000 import CHAIN_TAPS, chain1000_inputs, check_chain1000  # noqa: E402


def test_oracle_1000_step_chain_equals_reference():
    """infer.py:76-88 at --total_step 1000, cfg 9.0, B=2: the oracle (explicit softmax) against the reference run
    (fused SDPA) step by step and end to end."""
    xT, text, noises = chain1000_inputs()
    sd = synth.make_dit_state_dict(31337, gain=0.7)
    vsd = synth.make_vae_state_dict(2025)
    taps = {}
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():
        x = O.sample_ddpm(sd, xT, text, 1000, 9.0, noises,
                          on_step=lambda j, v: taps.__setitem__(j, v.clone().numpy()) if j in CHAIN_TAPS else None)
        series, _ = O.vae_decode(vsd, x, 96)
    check_chain1000(x.numpy(), series.numpy(), taps)
