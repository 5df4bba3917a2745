#!/usr/bin/env python3
"""Error of the two GPU arithmetics (f32 MFMA = the headline; bf16x3 = opt-in split-bf16 products) and of the fp32 CPU
oracle against an FP64 run of the oracle -- the evidence VERDICT r02 item 5 asks for before bf16x3 could ever become a
default.  Writes profiles/<tag>_accuracy.json and prints a markdown table.

"fp64 oracle" = the oracle's arithmetic in float64 on the SAME fp32 data: weights, text, noise, the DDPM schedule
tables and the time embedding (sin / cos of 100 t / f with arguments up to 1e5: their fp32 rounding is common to every
fp32 path and would swamp the table) are the fp32 values, upcast.

Cases: (i) one B = 8 forward (cond and uncond) at t in {0, 500, 999}; (ii) the 20-step DDPM and rectified-flow chains of
tests/golden/chains.npz; (iii) the 1000-step DDPM chain of tests/golden/chain1000.npz at its 7 taps + the decoded series.

    python tools/accuracy_table.py [--tag r03] [--skip-1000]
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracle import t2s_oracle as O  # noqa: E402
from t2ms_amd import synth  # noqa: E402


@contextlib.contextmanager
def fp64_arithmetic():
    """The oracle in float64 on fp32 data: default dtype float64; schedule tables and time embedding stay the fp32
    values (computed under a float32 default, then upcast)."""
    te, dt = O.time_embedding, O.ddpm_tables

    def in_f32(fn):
        def wrapped(*a, **k):
            torch.set_default_dtype(torch.float32)
            try:
                a = [x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a]
                r = fn(*a, **k)
            finally:
                torch.set_default_dtype(torch.float64)
            return {kk: v.double() for kk, v in r.items()} if isinstance(r, dict) else r.double()
        return wrapped

    O.time_embedding, O.ddpm_tables = in_f32(te), in_f32(dt)
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.set_default_dtype(torch.float32)
        O.time_embedding, O.ddpm_tables = te, dt


def dbl(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def err(x, ref):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(x - ref)
    scale = max(1.0, float(np.abs(ref).max()))
    return {"max_abs": float(d.max()), "rms": float(np.sqrt((d ** 2).mean())), "ref_max_abs": float(np.abs(ref).max()),
            "max_rel_to_ref_max": float(d.max()) / scale}


def gpu_model(sd, dev, math):
    from model.denoiser.transformer import Transformer
    m = Transformer()
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval().set_math(math)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--skip-1000", action="store_true")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    dev = torch.device("cuda", 0)
    import types
    from model.backbone.DDPM import DDPM
    from model.pretrained.vqvae import vqvae
    from t2ms_amd.sampler import Sampler
    vsd = synth.make_vae_state_dict(2025)
    vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    vae.load_state_dict(vsd, strict=True)
    vae = vae.to(dev).eval()
    table = []

    def record(case, ref, f32, x3, cpu32):
        row = {"case": case, "f32_mfma": err(f32, ref), "bf16x3": err(x3, ref), "cpu_fp32_oracle": err(cpu32, ref)}
        row["bf16x3_le_f32_mfma"] = bool(row["bf16x3"]["max_abs"] <= row["f32_mfma"]["max_abs"] and
                                         row["bf16x3"]["rms"] <= row["f32_mfma"]["rms"])
        table.append(row)
        print(f"{case}: f32 max {row['f32_mfma']['max_abs']:.3e} rms {row['f32_mfma']['rms']:.3e} | x3 max "
              f"{row['bf16x3']['max_abs']:.3e} rms {row['bf16x3']['rms']:.3e} | cpu32 max "
              f"{row['cpu_fp32_oracle']['max_abs']:.3e} rms {row['cpu_fp32_oracle']['rms']:.3e} | |ref| "
              f"{row['f32_mfma']['ref_max_abs']:.3g}", flush=True)

    # ---------------------------------------------------------------- (i) single forwards, B = 8
    sd = synth.make_dit_state_dict(2025)
    x = synth.make_latents(7, 8)
    text = synth.make_text_embeddings(7, 8)
    models = {k: gpu_model(sd, dev, k) for k in ("f32", "bf16x3")}
    for tval in (0, 500, 999):
        t = torch.full((8,), tval, dtype=torch.long)
        for name, tx in (("cond", text), ("uncond", None)):
            with torch.no_grad():
                cpu32 = O.dit_forward(sd, x, t, tx)
                with fp64_arithmetic():
                    ref = O.dit_forward(dbl(sd), x.double(), t, None if tx is None else tx.double())
                outs = {k: m(input=x.to(dev), t=t.to(dev), text_input=None if tx is None else tx.to(dev)).cpu()
                        for k, m in models.items()}
            record(f"forward B=8 t={tval} {name}", ref, outs["f32"], outs["bf16x3"], cpu32)

    # ---------------------------------------------------------------- (ii) the 20-step golden chains
    sd = synth.make_dit_state_dict(31337, gain=0.7)
    xT = synth.make_latents(31337, 4)
    text = synth.make_text_embeddings(31337, 4)
    noises = torch.from_numpy(np.random.RandomState(99).randn(20, 4, 64, 30).astype(np.float32))
    models = {k: gpu_model(sd, dev, k) for k in ("f32", "bf16x3")}
    with torch.no_grad():
        cpu32 = O.sample_ddpm(sd, xT, text, 20, 7.0, noises)
        with fp64_arithmetic():
            ref = O.sample_ddpm(dbl(sd), xT.double(), text.double(), 20, 7.0, noises.double())
    outs = {k: Sampler(m, vae.decoder, "ddpm", 20, 7.0, 4, 96, dev).run(text, x_T=xT, noise=noises)[0].cpu()
            for k, m in models.items()}
    record("20-step DDPM chain, cfg 7, B=4 (latent)", ref, outs["f32"], outs["bf16x3"], cpu32)
    with torch.no_grad():
        cpu32 = O.sample_rf(sd, xT, text, 20, 7.0)
        with fp64_arithmetic():
            ref = O.sample_rf(dbl(sd), xT.double(), text.double(), 20, 7.0)
    outs = {k: Sampler(m, vae.decoder, "flowmatching", 20, 7.0, 4, 96, dev).run(text, x_T=xT)[0].cpu()
            for k, m in models.items()}
    record("20-step rectified-flow chain, cfg 7, B=4 (latent)", ref, outs["f32"], outs["bf16x3"], cpu32)

    # ---------------------------------------------------------------- (iii) the 1000-step chain, 7 taps
    if not args.skip_1000:
        from _chain1000 import CHAIN_TAPS, chain1000_inputs
        xT, text, noises = chain1000_inputs()
        taps = {"cpu32": {}, "ref": {}}
        t0 = time.time()
        with torch.no_grad():
            fin32 = O.sample_ddpm(sd, xT, text, 1000, 9.0, noises,
                                  on_step=lambda j, x: taps["cpu32"].__setitem__(j, x.clone()) if j in CHAIN_TAPS else None)
            print(f"fp32 oracle chain: {time.time() - t0:.0f} s", flush=True)
            t0 = time.time()
            with fp64_arithmetic():
                fin64 = O.sample_ddpm(dbl(sd), xT.double(), text.double(), 1000, 9.0, noises.double(),
                                      on_step=lambda j, x: taps["ref"].__setitem__(j, x.clone()) if j in CHAIN_TAPS else None)
                ser64, _ = O.vae_decode(dbl(vsd), fin64, 96)
            print(f"fp64 oracle chain: {time.time() - t0:.0f} s", flush=True)
            ser32, _ = O.vae_decode(vsd, fin32, 96)
        gpu = {}
        for k, m in models.items():
            ddpm = DDPM(1000, dev)
            xg, textd, nz = xT.to(dev), text.to(dev), noises.to(dev)
            gt = {}
            with torch.no_grad():
                for j in range(1000):
                    t = torch.full((2,), 999 - j, dtype=torch.long, device=dev)
                    u = m(input=xg, t=t, text_input=None)
                    c = m(input=xg, t=t, text_input=textd)
                    xg = ddpm.p_sample(xg, u + 9.0 * (c - u), t, eps=nz[j])
                    if j in CHAIN_TAPS:
                        gt[j] = xg.cpu()
                ser, _ = vae.decoder(xg, length=96)
            fused = Sampler(m, vae.decoder, "ddpm", 1000, 9.0, 2, 96, dev).run(text, x_T=xT, noise=noises)
            gpu[k] = dict(taps=gt, series=ser.cpu(), fused_latent=fused[0].cpu(), fused_series=fused[1].cpu())
        for j in CHAIN_TAPS:
            record(f"1000-step DDPM chain, cfg 9, B=2: x after loop index {j}", taps["ref"][j], gpu["f32"]["taps"][j],
                   gpu["bf16x3"]["taps"][j], taps["cpu32"][j])
        record("1000-step chain: fused sampler final latent", fin64, gpu["f32"]["fused_latent"],
               gpu["bf16x3"]["fused_latent"], fin32)
        record("1000-step chain: decoded series (B,96)", ser64, gpu["f32"]["fused_series"], gpu["bf16x3"]["fused_series"],
               ser32)

    n_le = sum(r["bf16x3_le_f32_mfma"] for r in table)
    out = {"reference": "fp64 arithmetic of oracle/t2s_oracle.py on the fp32 weights / inputs / schedule / time embedding",
           "rows": table, "bf16x3_le_f32_mfma_everywhere": n_le == len(table), "entries": len(table),
           "entries_where_bf16x3_le_f32_mfma": n_le}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    path = os.path.join(REPO, "gpurun_out", f"{args.tag}_accuracy.json")
    json.dump(out, open(path, "w"), indent=1)
    print("\n| case | |ref| max | f32 MFMA max / rms | bf16x3 max / rms | CPU fp32 oracle max / rms |")
    print("|---|---|---|---|---|")
    for r in table:
        print(f"| {r['case']} | {r['f32_mfma']['ref_max_abs']:.3g} | {r['f32_mfma']['max_abs']:.2e} / {r['f32_mfma']['rms']:.2e} | "
              f"{r['bf16x3']['max_abs']:.2e} / {r['bf16x3']['rms']:.2e} | {r['cpu_fp32_oracle']['max_abs']:.2e} / "
              f"{r['cpu_fp32_oracle']['rms']:.2e} |")
    print(f"\nbf16x3 <= f32 MFMA (max and rms) in {n_le} of {len(table)} entries; wrote {path}")


if __name__ == "__main__":
    main()
