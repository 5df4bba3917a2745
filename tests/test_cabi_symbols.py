"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly the
symbols include/t2s.h declares; the host mirrors keep the reference's state-dict layout."""
import os
import re
import types

import pytest
import torch

from t2ms_amd import _lib as L
from t2ms_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(REPO, "include", "t2s.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(t2s_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    lib = L.lib()  # raises if the .so is missing or lacks a symbol
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.t2s_version()


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(L.DitBlockWeights) == 10 * 8
    assert C.sizeof(L.DitWeights) == 10 * 8 + 4 * 10 * 8
    assert C.sizeof(L.VaeStackWeights) == 8 * 8
    assert C.sizeof(L.VaeWeights) == 16 + 8 * (6 + 8) + 2 * 64
    assert C.sizeof(L.SampleConfig) == 56


def test_state_dict_layout_is_the_references():
    from model.denoiser.transformer import Transformer
    from model.pretrained.vqvae import vqvae
    m = Transformer()
    sd = m.state_dict()
    ref = synth.make_dit_state_dict(3)
    assert set(sd) == set(ref)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in m.parameters()) == 1007705
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 946265
    # a15: adaLN output layer zero-initialised, linear biases zero
    for blk in m.layers:
        assert float(blk.adaLN_modulation[-1].weight.abs().max()) == 0.0
        assert float(blk.attn.qkv.bias.abs().max()) == 0.0
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                    embedding_dim=64))
    assert set(v.state_dict()) == set(synth.make_vae_state_dict(3))
    assert sum(p.numel() for p in v.parameters()) == 672833
    # the encoder can be grafted onto the DiT like infer.py:47 / train.py:30 and its keys appear
    m.encoder = v.encoder
    assert sum(1 for k in m.state_dict() if k.startswith("encoder.")) == 12


def test_product_fails_loudly_on_cpu_tensors():
    from model.denoiser.transformer import Transformer
    from model.backbone.DDPM import DDPM
    m = Transformer()
    with torch.no_grad(), pytest.raises(L.T2SError, match="GPU"):
        m(input=torch.zeros(2, 64, 30), t=torch.zeros(2, dtype=torch.long), text_input=None)
    d = DDPM(10, "cpu")
    with pytest.raises(L.T2SError, match="GPU"):
        d.p_sample(torch.zeros(2, 64, 30), torch.zeros(2, 64, 30), torch.zeros(2, dtype=torch.long))


def test_pickle_roundtrip_uses_reference_module_paths(tmp_path):
    from model.pretrained.vqvae import vqvae
    v = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256,
                                    embedding_dim=64))
    p = tmp_path / "final_model.pth"
    torch.save(v, p)
    raw = open(p, "rb").read()
    assert b"model.pretrained.vqvae" in raw and b"t2ms_amd.model.pretrained.vqvae" not in raw
    w = torch.load(p, map_location="cpu", weights_only=False)
    assert type(w).__name__ == "vqvae" and hasattr(w, "encoder") and hasattr(w, "decoder")
