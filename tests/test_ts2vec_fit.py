"""TS2Vec.fit (reference evaluate/ts2vec.py:73-160, trained by evaluation.py:238 for C-FID) -- SURVEY.md 8f row 4.
tests/golden/ts2vec_fit.npz is the REFERENCE's run (tests/golden/gen_golden_r3.py): 12 iterations under seeds 7 and the
default 200-iteration run under seeds 8 on 24 seeded series.  t2ms_amd.ts2vec follows the reference's order and sources
of random draws, so on the CPU (torch autograd both sides) the loss curve is reproduced exactly; on the GPU the training
arithmetic differs in the last bits and the HIP encoder kernel produces the representations."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ts2vec_fit.npz")


def _fit(device, seed, n_iters):
    from t2ms_amd.ts2vec import TS2Vec
    g = np.load(GOLD)
    torch.manual_seed(seed)
    np.random.seed(seed)
    losses = []
    m = TS2Vec(input_dims=1, device=device, batch_size=8, lr=0.001, output_dims=100, max_train_length=3000,
               after_iter_callback=lambda model, loss: losses.append(loss))
    log = m.fit(g["ori"].copy(), n_iters=n_iters, verbose=False)
    return g, m, np.asarray(losses), log


def test_fit_reproduces_the_reference_loss_curve_on_cpu():
    g, m, losses, log = _fit("cpu", 7, 12)
    assert m.n_iters == 12 and len(losses) == 12
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(np.asarray(log), g["epoch_log"], rtol=1e-6)
    # state-dict layout of the averaged encoder = the reference's (what TS2VecEncoder / save() / load() exchange)
    keys = set(m.net.state_dict())
    assert "n_averaged" in keys and "module.input_fc.weight" in keys and "module.feature_extractor.net.10.projector.weight" in keys
    assert sum(p.numel() for p in m._net.parameters()) == sum(v.numel() for k, v in m.net.state_dict().items() if k != "n_averaged")
    with pytest.raises(Exception, match="GPU"):
        m.encode(g["ori"], encoding_window="full_series")          # the encoder kernel has no CPU fallback


@pytest.mark.gpu
def test_fit_on_gpu_then_hip_encoder_and_cfid():
    from t2ms_amd import metrics
    assert torch.cuda.is_available()
    g, m, losses, _ = _fit("cuda:0", 7, 12)
    # same draws, GPU arithmetic: the curve follows the reference's within the drift of 12 optimisation steps
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-3)
    r_ori = m.encode(g["ori"], encoding_window="full_series")
    r_gen = m.encode(g["gen"], encoding_window="full_series")
    assert r_ori.shape == (24, 100)
    scale = float(np.abs(g["repr_ori"]).max())
    assert float(np.abs(r_ori - g["repr_ori"]).max()) < 2e-2 * scale and float(np.abs(r_gen - g["repr_gen"]).max()) < 2e-2 * scale
    # the representations come from the HIP kernel on the averaged weights: equal to torch's eval forward of those weights
    fid_ref = metrics.fid(g["repr_ori"], g["repr_gen"])
    fid_ours = metrics.fid(r_ori, r_gen)
    assert abs(fid_ours - fid_ref) <= 0.05 * abs(fid_ref) + 1e-3, (fid_ours, fid_ref)


@pytest.mark.gpu
def test_default_200_iteration_fit_as_evaluation_py_runs_it():
    from t2ms_amd.ts2vec import initialize_ts2vec
    g = np.load(GOLD)
    torch.manual_seed(8)
    np.random.seed(8)
    log = []
    from t2ms_amd import ts2vec as T
    m = T.TS2Vec(input_dims=1, device="cuda:0", batch_size=8, lr=0.001, output_dims=100, max_train_length=3000,
                 after_iter_callback=lambda model, loss: log.append(loss))
    m.fit(g["ori"].copy(), verbose=False)
    assert [m.n_iters, m.n_epochs] == g["n_iters200"].tolist()      # 200 iterations (data.size <= 100000), 66 epochs of 3
    log = np.asarray(log)
    np.testing.assert_allclose(log[:10], g["losses200"][:10], rtol=5e-3)
    assert abs(log[-30:].mean() - g["losses200"][-30:].mean()) < 0.15 * g["losses200"][-30:].mean()
    assert log[-30:].mean() < 0.5 * log[:5].mean()
    model = initialize_ts2vec(g["ori"].copy(), device="cuda:0")       # the evaluation.py:238 entry point
    assert model.n_iters == 200 and np.isfinite(model.encode(g["gen"], encoding_window="full_series")).all()
