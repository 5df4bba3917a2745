#!/usr/bin/env python3
"""Round-2 fixtures, again made by RUNNING THE REFERENCE in the build container (see gen_golden.py for the rules:
the reference never travels, only the arrays written here are committed; `timm` is supplied by gen_golden's
stand-in, `dtaidistance` -- imported at evaluation.py:7, not installed -- by an import-time stub whose function is
never called by anything recorded below).

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/gen_golden_r2.py

Writes
  init_seeded.npz   Transformer() under torch.manual_seed(k): parameter ORDER and per-tensor statistics + strided
                    samples of the initialised values (transformer.py:128-154,194-204)           -> SURVEY 8a row a15
  dataset.npz + dataset_csv/*.csv   two tiny synthetic CSVs (both TextEmbedding formats, a constant OT column) and
                    what the reference T2SDataset (datafactory/dataset.py:10-104) makes of them, train and test
                                                                                                   -> SURVEY 8f row 3
  metrics.npz       calculate_mse / _wape / _mrr / _crps / _ed / calculate_fid (evaluation.py) and cosine_similarity
                    (Dataset_Construction_Pipeline/Evaluate_Datasets.py:6-15) on seeded arrays     -> SURVEY 8f row 4
  chain1000.npz     a B=2, T=1000 DDPM chain at cfg 9.0 through the reference loop (infer.py:76-88), final latent,
                    decoded series and a few tapped steps                                          -> north_star 1e-4
  ts2vec.npz        TSEncoder forward of evaluate/ts2vec.py (:352-399) on seeded weights, per-step and
                    full-series max-pooled representations                                         -> SURVEY 8f row 4
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def save(name, **arrs):
    arrs = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def write_dataset_csvs(root):
    """Our own synthetic rows in the reference's CSV contract (dataset.py:71-96)."""
    import pandas as pd
    os.makedirs(root, exist_ok=True)
    rs = np.random.RandomState(314)
    n, L = 60, 24
    series = np.round(rs.uniform(-3, 7, size=(n, L)), 4)
    series[3] = series[3] * 10                            # one row that owns several column maxima
    series[:, 5] = 2.5                                   # a constant column: MinMaxScaler maps it to 0
    emb = rs.randn(n, 128)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    texts = [f"series {i} rises then falls, \"quoted\", with comma" for i in range(n)]
    # TSFragment family: whitespace-separated floats in brackets over several lines (numpy's repr of an array)
    frag = pd.DataFrame({"Text": texts, "OT": [str(list(map(float, r))) for r in series],
                         "TextEmbedding": [np.array2string(e, precision=8, max_line_width=75) for e in emb]})
    frag.to_csv(os.path.join(root, "embedding_cleaned_ETTh1_24.csv"), index=False)
    # MMD family ("Climate" in the name): a python-literal list
    mmd = pd.DataFrame({"Text": texts, "OT": [str(list(map(float, r))) for r in series],
                        "TextEmbedding": [str([float(f"{v:.8f}") for v in e]) for e in emb]})
    mmd.to_csv(os.path.join(root, "embedding_cleaned_Climate_24.csv"), index=False)


def main():
    torch.set_num_threads(8)
    from gen_golden import _install_timm_stub
    from t2ms_amd import synth
    _install_timm_stub()
    dt = types.ModuleType("dtaidistance")
    dn = types.ModuleType("dtaidistance.dtw_ndim")
    dn.distance = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("dtaidistance is not installed"))
    dt.dtw_ndim = dn
    sys.modules.update({"dtaidistance": dt, "dtaidistance.dtw_ndim": dn})
    # The repo's own `model/` and `datafactory/` are REGULAR packages (they hold the drop-in mirrors) and would shadow the
    # reference's namespace packages of the same name under any sys.path order: take the repo off the path now that
    # t2ms_amd.synth is imported, and check below that what was imported IS the reference.
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    for k in [k for k in sys.modules if k.split(".")[0] in ("model", "datafactory", "evaluate", "evaluation")]:
        del sys.modules[k]
    os.chdir(HERE)                      # '' on sys.path must not resolve to the repo root either
    sys.path.insert(0, REF)
    from model.denoiser.transformer import Transformer
    from model.backbone.DDPM import DDPM
    from model.pretrained.vqvae import vqvae
    for mod in ("model.denoiser.transformer", "model.backbone.DDPM", "model.pretrained.vqvae"):
        assert sys.modules[mod].__file__.startswith(REF + os.sep), (mod, sys.modules[mod].__file__)

    # (10) seeded construction --------------------------------------------------------------------------------
    out = {}
    for seed in (0, 2025):
        torch.manual_seed(seed)
        m = Transformer()
        names = [n for n, _ in m.named_parameters()]
        out[f"names_{seed}"] = np.asarray(names)
        for n, p in m.named_parameters():
            v = p.detach().double().flatten()
            key = n.replace(".", "__")
            out[f"stat_{seed}_{key}"] = np.asarray([float(v.sum()), float((v * v).sum()), float(v.abs().max())])
            out[f"samp_{seed}_{key}"] = p.detach().flatten()[::97].clone()
        out[f"rng_after_{seed}"] = torch.rand(4)          # the generator state the constructor leaves behind
    save("init_seeded", **out)

    # (11) dataset ----------------------------------------------------------------------------------------------
    csv_root = os.path.join(HERE, "dataset_csv")
    write_dataset_csvs(csv_root)
    from datafactory.dataset import T2SDataset
    assert sys.modules["datafactory.dataset"].__file__.startswith(REF + os.sep)
    out = {}
    np.random.seed(99)
    state_before = np.random.get_state()[1][:4].copy()
    for name in ("embedding_cleaned_ETTh1_24", "embedding_cleaned_Climate_24"):
        for period in ("train", "test"):
            ds = T2SDataset(name=name, data_root=csv_root, period=period, proportion=0.9)
            key = f"{name.split('_')[2]}_{period}"
            out[f"samples_{key}"] = np.asarray(ds.samples, dtype=np.float64)
            out[f"embedding_{key}"] = np.asarray(ds.embedding, dtype=np.float64)
            out[f"text_{key}"] = np.asarray(ds.text)
            out[f"meta_{key}"] = np.asarray([len(ds), ds.len, ds.var_num])
            t0, x0, e0 = ds[1]
            out[f"item1_x_{key}"], out[f"item1_e_{key}"], out[f"item1_t_{key}"] = np.asarray(x0), np.asarray(e0), np.asarray(t0)
    # the default 99 % split of 60 rows: ceil(59.4) = 60 train rows, an EMPTY test split
    out["default_split_lens"] = np.asarray([len(T2SDataset(name="embedding_cleaned_ETTh1_24", data_root=csv_root, period=p))
                                            for p in ("train", "test")])
    assert np.array_equal(np.random.get_state()[1][:4], state_before)   # divide() restores the global RNG
    save("dataset", **out)

    # (12) metrics ----------------------------------------------------------------------------------------------
    import evaluation as E
    from Dataset_Construction_Pipeline.Evaluate_Datasets import cosine_similarity
    assert E.__file__.startswith(REF + os.sep)
    rs = np.random.RandomState(2718)
    N, L, R = 9, 24, 10
    ori = rs.uniform(0, 1, size=(N, L, 1)).astype(np.float32)           # the (N, L, 1) arrays infer.py writes
    gen = (ori + 0.2 * rs.randn(N, L, 1)).astype(np.float32)
    ori[4] = 0.0                                                          # all-zero row: WAPE nan, cosine nan -> 0
    runs = np.stack([(ori + s * rs.randn(N, L, 1)).astype(np.float32) for s in np.linspace(0.05, 1.5, R)], axis=-1)
    runs[2] = -runs[2]                                                    # negative similarity everywhere: score 0
    runs[6, :, :, 7] = ori[6]                                             # an exact copy at run index 7
    # evaluate_data transposes to (N, 1, L) before MSE / WAPE (evaluation.py:246-247 of the __main__ + :235)
    o_t, g_t = np.transpose(ori, (0, 2, 1)), np.transpose(gen, (0, 2, 1))
    o_tt, g_tt = np.transpose(o_t, (0, 2, 1)), np.transpose(g_t, (0, 2, 1))
    E.therehold = 0.5                                                     # module global set in __main__ (:299)
    sims = np.asarray([[np.mean(cosine_similarity(ori[i], runs[i, :, :, g])) for g in range(R)] for i in range(N)])
    out = dict(ori=ori, gen=gen, runs=runs,
               mse=E.calculate_mse(o_tt, g_tt), wape=E.calculate_wape(o_tt, g_tt), mrr=E.calculate_mrr(ori, runs),
               crps=E.calculate_crps(ori, runs), ed=E.calculate_ed(ori, gen), sims=sims,
               cos_pairs=np.asarray([cosine_similarity([1, 0, 0], [1, 1, 0]), cosine_similarity([0, 0], [1, 1]),
                                     cosine_similarity(ori[0], gen[0])]))
    a1 = rs.randn(40, 6) @ rs.randn(6, 6)
    a2 = rs.randn(40, 6) @ rs.randn(6, 6) + 0.3
    out.update(fid_act1=a1, fid_act2=a2, fid=E.calculate_fid(a1, a2))
    save("metrics", **out)

    # (13) 1000-step DDPM chain at the headline schedule ----------------------------------------------------------
    sdc = synth.make_dit_state_dict(31337, gain=0.7)        # contractive weights: state stays O(1)
    model = Transformer().eval()
    model.load_state_dict(sdc, strict=True)
    ns = types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64)
    vae = vqvae(ns).eval()
    vae.load_state_dict(synth.make_vae_state_dict(2025), strict=True)
    B, steps, cfg = 2, 1000, 9.0
    xT = synth.make_latents(1000, B)
    text = synth.make_text_embeddings(1000, B)
    ddpm = DDPM(steps, "cpu")
    taps = {}
    tap_at = (0, 1, 9, 99, 499, 998, 999)
    with torch.no_grad():
        x = xT.clone()
        for j in range(steps):
            tt = torch.full((B,), steps - 1 - j, dtype=torch.long)
            u = model(input=x, t=tt, text_input=None)
            c = model(input=x, t=tt, text_input=text)
            pred = u + cfg * (c - u)
            torch.manual_seed(50_000 + j)                   # p_sample draws torch.randn(xt.shape) (DDPM.py:35)
            x = ddpm.p_sample(x, pred, tt)
            if j in tap_at:
                taps[f"x_after_{j}"] = x.clone()
        series, _ = vae.decoder(x, length=96)
    save("chain1000", latent=x, series=series, max_abs=np.asarray(float(x.abs().max())), **taps)

    # (14) TS2Vec encoder forward ----------------------------------------------------------------------------------
    from evaluate.ts2vec import TSEncoder
    assert sys.modules["evaluate.ts2vec"].__file__.startswith(REF + os.sep)
    enc = TSEncoder(input_dims=1, output_dims=100, hidden_dims=64, depth=10).eval()   # initialize_ts2vec's sizes (:12-19)
    print("TSEncoder load_state_dict(strict):", enc.load_state_dict(synth.make_ts2vec_state_dict(2025), strict=True))
    xs = torch.from_numpy(np.random.RandomState(8).uniform(0, 1, size=(5, 96, 1)).astype(np.float32))
    xs[1, 10:14, 0] = float("nan")                          # TSEncoder zeroes time steps that hold a NaN (:367-368)
    with torch.no_grad():
        rep = enc(xs.clone(), mask="all_true")              # (5, 96, 100): what encode() pools (:236-245)
        full = torch.nn.functional.max_pool1d(rep.transpose(1, 2), kernel_size=rep.size(1)).squeeze(-1)
    save("ts2vec", x=xs, rep=rep[:, ::6], full_series=full)


if __name__ == "__main__":
    main()
