#!/usr/bin/env python3
"""Is the bf16x3 arithmetic (fp32 operands split exactly into three bf16 terms, six bf16 MFMAs per product, fp32 accumulate)
"not narrower than the reference's own arithmetic"?  Error of the three fp32 arithmetics

    f32 MFMA (v_mfma_f32_32x32x2_f32)   |   bf16x3   |   the CPU fp32 oracle (= the reference's PyTorch-CPU arithmetic)

against an FP64 run of the oracle, at a sample that can decide: B = 256 rows x 3 seeds for single forwards
and for the 20-step DDPM / rectified-flow chains, B = 32 for the taps of the 1000-step chain at the headline schedule.
Per entry: rms and max error of each column against fp64, the ratios bf16x3 / oracle, and the bar

    rms(bf16x3) <= 1.05 x rms(CPU fp32 oracle)   and   max(bf16x3) <= 1.25 x max(CPU fp32 oracle)        (every entry).

"fp64 oracle" = oracle/t2s_oracle.py with float64 arithmetic on the SAME fp32 data: weights, text, noise, the DDPM schedule
tables and the time embedding (sin / cos of 100 t / f with arguments up to 1e5: their fp32 rounding is common to every fp32
path and would swamp the table) are the fp32 values, upcast.

Two stages, because the CPU references cost ~1 h of host time and a GPU box is metered:

    python tools/accuracy_table.py --stage cpu              # anywhere: writes tools/_acc_ref/*.npz (git-ignored, travels with gpurun)
    python tools/accuracy_table.py --stage gpu --tag r05    # on the GPU box: the two GPU arithmetics against those files
                                                            # -> gpurun_out/<tag>_accuracy.{json,md}

`--small` is the round-3 sample (B = 8 / 4 / 2) for a quick look; `--stage all` does both in one process.
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

from oracle import t2s_oracle as O  # noqa: E402   (the checker: this tool measures the product against it)
from t2ms_amd import synth  # noqa: E402

REF_DIR = os.path.join(REPO, "tools", "_acc_ref")
TAPS = (0, 1, 9, 99, 499, 998, 999)
RMS_BAR, MAX_BAR = 1.05, 1.25


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


# ------------------------------------------------------------------------------------------------ the cases (shared by both stages)
def plan(small: bool):
    B_fwd, B_chain, B_long, seeds = (8, 4, 2, (0,)) if small else (256, 256, 32, (0, 1, 2))
    return dict(B_fwd=B_fwd, B_chain=B_chain, B_long=B_long, seeds=seeds)


def fwd_inputs(seed, B):
    return synth.make_dit_state_dict(2025 + seed), synth.make_latents(7 + 100 * seed, B), synth.make_text_embeddings(7 + 100 * seed, B)


def chain_inputs(seed, B):
    sd = synth.make_dit_state_dict(31337 + seed, gain=0.7)
    xT, text = synth.make_latents(31337 + seed, B), synth.make_text_embeddings(31337 + seed, B)
    noises = torch.from_numpy(np.random.RandomState(99 + seed).randn(20, B, 64, 30).astype(np.float32))
    return sd, xT, text, noises


def long_inputs(B):
    """1000-step DDPM at the headline schedule (cfg 9), the weights of tests/golden/chain1000.npz, B rows."""
    sd = synth.make_dit_state_dict(31337, gain=0.7)
    xT, text = synth.make_latents(1000, B), synth.make_text_embeddings(1000, B)
    noises = torch.empty(1000, B, 64, 30)
    for j in range(1000):
        noises[j] = torch.from_numpy(np.random.RandomState(70_000 + j).randn(B, 64, 30).astype(np.float32))
    return sd, xT, text, noises


def ref_path(name, small):
    return os.path.join(REF_DIR, ("small_" if small else "") + name + ".npz")


def host_tables():
    """Inputs that are EVALUATED ON THE HOST and therefore depend on the machine's libm / SIMD paths (torch.sin / torch.exp of
    the positional table, transformer.py:14-23; torch.pow of the time frequencies, :34): an ulp of difference between the box
    that made the references and the box that runs the GPU columns is an INPUT difference of ~1e-7 relative -- it showed up as
    a uniform 4x error of both GPU columns when the two stages first ran on different machines.  Stage cpu records them,
    stage gpu feeds exactly these to the kernels (and reports how far its own host values are from them)."""
    from t2ms_amd.model.backbone.DDPM import ddpm_host_tables
    out = {"pos_embed": synth.make_dit_state_dict(2025)["pos_embed"].numpy(), "time_freqs": O.time_freqs().numpy()}
    # the DDPM schedule and the per-step coefficients the kernels read (DDPM.py:14-18,30-36: torch.linspace / cumprod / pow
    # on the host): an ulp of difference in a coefficient is a multiplicative perturbation of the state at that step, and over
    # 1000 steps of a state that grows to 1.5e3 it showed up as 4 x the error of BOTH GPU arithmetics at the late taps
    for T in (20, 1000):
        for k, v in ddpm_host_tables(T).items():
            out[f"ddpm_{T}_{k}"] = v.numpy()
    return out


# ------------------------------------------------------------------------------------------------ stage cpu
def stage_cpu(args):
    p = plan(args.small)
    os.makedirs(REF_DIR, exist_ok=True)
    torch.set_num_threads(args.threads)

    def have(name):
        return os.path.exists(ref_path(name, args.small)) and not args.force

    def save(name, **arrs):
        np.savez(ref_path(name, args.small), **arrs)
        print(f"[cpu] wrote {ref_path(name, args.small)}", flush=True)

    save("host_tables", **host_tables())

    with torch.no_grad():
        for seed in p["seeds"]:
            name = f"fwd_s{seed}"
            if have(name):
                continue
            sd, x, text = fwd_inputs(seed, p["B_fwd"])
            out = {}
            t0 = time.time()
            for tval in (0, 500, 999):
                t = torch.full((p["B_fwd"],), tval, dtype=torch.long)
                for nm, tx in (("cond", text), ("uncond", None)):
                    out[f"cpu32_t{tval}_{nm}"] = O.dit_forward(sd, x, t, tx).numpy()
                    with fp64_arithmetic():
                        out[f"ref_t{tval}_{nm}"] = O.dit_forward(dbl(sd), x.double(), t, None if tx is None else tx.double()).numpy()
            print(f"[cpu] forwards seed {seed}: {time.time() - t0:.0f} s", flush=True)
            save(name, **out)
        for seed in p["seeds"]:
            name = f"chain20_s{seed}"
            if have(name):
                continue
            sd, xT, text, noises = chain_inputs(seed, p["B_chain"])
            t0 = time.time()
            out = {"cpu32_ddpm": O.sample_ddpm(sd, xT, text, 20, 7.0, noises).numpy(),
                   "cpu32_rf": O.sample_rf(sd, xT, text, 20, 7.0).numpy()}
            with fp64_arithmetic():
                out["ref_ddpm"] = O.sample_ddpm(dbl(sd), xT.double(), text.double(), 20, 7.0, noises.double()).numpy()
                out["ref_rf"] = O.sample_rf(dbl(sd), xT.double(), text.double(), 20, 7.0).numpy()
            print(f"[cpu] 20-step chains seed {seed}: {time.time() - t0:.0f} s", flush=True)
            save(name, **out)
        if not args.skip_1000 and not have("chain1000"):
            sd, xT, text, noises = long_inputs(p["B_long"])
            vsd = synth.make_vae_state_dict(2025)
            out = {}
            t0 = time.time()

            def tap(prefix):
                def f(j, x):
                    if j in TAPS:
                        out[f"{prefix}_x{j}"] = x.clone().numpy()
                    if j % 100 == 99:
                        print(f"[cpu] 1000-step {prefix}: step {j + 1}, {time.time() - t0:.0f} s", flush=True)
                return f
            fin32 = O.sample_ddpm(sd, xT, text, 1000, 9.0, noises, on_step=tap("cpu32"))
            out["cpu32_series"] = O.vae_decode(vsd, fin32, 96)[0].numpy()
            with fp64_arithmetic():
                fin64 = O.sample_ddpm(dbl(sd), xT.double(), text.double(), 1000, 9.0, noises.double(), on_step=tap("ref"))
                out["ref_series"] = O.vae_decode(dbl(vsd), fin64, 96)[0].numpy()
            save("chain1000", **out)


# ------------------------------------------------------------------------------------------------ stage gpu
def err(x, ref):
    d = np.abs(np.asarray(x, dtype=np.float64) - np.asarray(ref, dtype=np.float64))
    return {"max_abs": float(d.max()), "rms": float(np.sqrt((d ** 2).mean())), "ref_max_abs": float(np.abs(ref).max()), "n": int(d.size)}


HOST = {}


def use_reference_box_schedule():
    """Make the mirrors' DDPM (class API) and the fused Sampler read the reference box's schedule tables."""
    import t2ms_amd.model.backbone.DDPM as D
    import t2ms_amd.sampler as S
    real = D.ddpm_host_tables

    def tables(total_steps):
        if f"ddpm_{total_steps}_coef" not in HOST:
            return real(total_steps)
        return {k: torch.from_numpy(HOST[f"ddpm_{total_steps}_{k}"]) for k in ("beta", "alpha", "alpha_bar", "coef", "sqrt_ab", "sqrt_1mab")}
    D.ddpm_host_tables = tables
    S.ddpm_host_tables = tables


def gpu_model(sd, dev, math):
    import t2ms_amd.model.denoiser.transformer as T       # (model.denoiser.transformer is an alias package of this module)
    sd = dict(sd)
    if HOST:          # the reference box's host-evaluated tables (host_tables)
        sd["pos_embed"] = torch.from_numpy(HOST["pos_embed"])
        T._FREQS[str(torch.device(dev))] = torch.from_numpy(HOST["time_freqs"]).to(dev)
    m = T.Transformer()
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval().set_math(math)


def stage_gpu(args):
    import types

    from model.pretrained.vqvae import vqvae
    from t2ms_amd.sampler import Sampler
    p = plan(args.small)
    dev = torch.device("cuda", 0)
    vsd = synth.make_vae_state_dict(2025)
    vae = vqvae(types.SimpleNamespace(block_hidden_size=128, num_residual_layers=2, res_hidden_size=256, embedding_dim=64))
    vae.load_state_dict(vsd, strict=True)
    vae = vae.to(dev).eval()
    table = []
    MATHS = ("f32", "bf16x3")
    host_diff = None
    if os.path.exists(ref_path("host_tables", args.small)):
        HOST.update({k: v for k, v in np.load(ref_path("host_tables", args.small)).items()})
        mine = host_tables()
        host_diff = {k: {"max_abs_diff": float(np.abs(mine[k].astype(np.float64) - HOST[k]).max()),
                         "max_rel_diff": float((np.abs(mine[k].astype(np.float64) - HOST[k]) / np.maximum(np.abs(HOST[k]), 1e-30)).max()),
                         "elements_differing": int((mine[k] != HOST[k]).sum()), "elements": int(HOST[k].size)} for k in HOST if k in mine}
        print("host-evaluated tables, this box vs the reference box:", {k: v for k, v in host_diff.items() if v["elements_differing"]}, flush=True)
        use_reference_box_schedule()

    def record(case, ref, f32, x3, cpu32):
        cat = lambda xs: np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in xs])   # noqa: E731  (pooled over seeds)
        ref, f32, x3, cpu32 = cat(ref), cat(f32), cat(x3), cat(cpu32)
        row = {"case": case, "f32_mfma": err(f32, ref), "bf16x3": err(x3, ref), "cpu_fp32_oracle": err(cpu32, ref)}
        row["x3_over_oracle_rms"] = row["bf16x3"]["rms"] / row["cpu_fp32_oracle"]["rms"]
        row["x3_over_oracle_max"] = row["bf16x3"]["max_abs"] / row["cpu_fp32_oracle"]["max_abs"]
        row["f32_over_oracle_rms"] = row["f32_mfma"]["rms"] / row["cpu_fp32_oracle"]["rms"]
        row["f32_over_oracle_max"] = row["f32_mfma"]["max_abs"] / row["cpu_fp32_oracle"]["max_abs"]
        row["x3_meets_bar"] = bool(row["x3_over_oracle_rms"] <= RMS_BAR and row["x3_over_oracle_max"] <= MAX_BAR)
        row["f32_meets_bar"] = bool(row["f32_over_oracle_rms"] <= RMS_BAR and row["f32_over_oracle_max"] <= MAX_BAR)
        table.append(row)
        print(f"{case}: x3/oracle rms {row['x3_over_oracle_rms']:.3f} max {row['x3_over_oracle_max']:.3f} | f32/oracle rms "
              f"{row['f32_over_oracle_rms']:.3f} max {row['f32_over_oracle_max']:.3f} | oracle rms {row['cpu_fp32_oracle']['rms']:.3e} max "
              f"{row['cpu_fp32_oracle']['max_abs']:.3e} | n {row['bf16x3']['n']}", flush=True)

    # (i) single forwards
    acc = {}
    for seed in p["seeds"]:
        g = np.load(ref_path(f"fwd_s{seed}", args.small))
        sd, x, text = fwd_inputs(seed, p["B_fwd"])
        models = {k: gpu_model(sd, dev, k) for k in MATHS}
        for tval in (0, 500, 999):
            t = torch.full((p["B_fwd"],), tval, dtype=torch.long)
            for nm, tx in (("cond", text), ("uncond", None)):
                with torch.no_grad():
                    outs = {k: m(input=x.to(dev), t=t.to(dev), text_input=None if tx is None else tx.to(dev)).cpu().numpy()
                            for k, m in models.items()}
                a = acc.setdefault((tval, nm), ([], [], [], []))
                a[0].append(g[f"ref_t{tval}_{nm}"]); a[1].append(outs["f32"]); a[2].append(outs["bf16x3"]); a[3].append(g[f"cpu32_t{tval}_{nm}"])
        del models
    for (tval, nm), a in acc.items():
        record(f"forward, B={p['B_fwd']} x {len(p['seeds'])} seeds, t={tval}, {nm}", *a)

    # (ii) 20-step chains
    acc = {"ddpm": ([], [], [], []), "rf": ([], [], [], [])}
    for seed in p["seeds"]:
        g = np.load(ref_path(f"chain20_s{seed}", args.small))
        sd, xT, text, noises = chain_inputs(seed, p["B_chain"])
        for kind, backbone in (("ddpm", "ddpm"), ("rf", "flowmatching")):
            outs = {}
            for k in MATHS:
                m = gpu_model(sd, dev, k)
                s = Sampler(m, vae.decoder, backbone, 20, 7.0, p["B_chain"], 96, dev)
                outs[k] = s.run(text, x_T=xT, noise=noises if kind == "ddpm" else None)[0].cpu().numpy()
                del s, m
            a = acc[kind]
            a[0].append(g[f"ref_{kind}"]); a[1].append(outs["f32"]); a[2].append(outs["bf16x3"]); a[3].append(g[f"cpu32_{kind}"])
    record(f"20-step DDPM chain, cfg 7, B={p['B_chain']} x {len(p['seeds'])} seeds (latent)", *acc["ddpm"])
    record(f"20-step rectified-flow chain, cfg 7, B={p['B_chain']} x {len(p['seeds'])} seeds (latent)", *acc["rf"])

    # (iii) the 1000-step chain at the headline schedule: class-API loop for the taps (the reference's own loop shape,
    # infer.py:76-88), fused sampler for the final latent and the decoded series
    if not args.skip_1000 and os.path.exists(ref_path("chain1000", args.small)):
        from model.backbone.DDPM import DDPM
        g = np.load(ref_path("chain1000", args.small))
        B = p["B_long"]
        sd, xT, text, noises = long_inputs(B)
        gpu = {}
        for k in MATHS:
            m = gpu_model(sd, dev, k)
            ddpm = DDPM(1000, dev)
            xg, textd, nz = xT.to(dev), text.to(dev), noises.to(dev)
            gt = {}
            with torch.no_grad():
                for j in range(1000):
                    t = torch.full((B,), 999 - j, dtype=torch.long, device=dev)
                    u = m(input=xg, t=t, text_input=None)
                    c = m(input=xg, t=t, text_input=textd)
                    xg = ddpm.p_sample(xg, u + 9.0 * (c - u), t, eps=nz[j])
                    if j in TAPS:
                        gt[j] = xg.cpu().numpy()
            fused = Sampler(m, vae.decoder, "ddpm", 1000, 9.0, B, 96, dev).run(text, x_T=xT, noise=noises)
            gpu[k] = dict(taps=gt, fused_latent=fused[0].cpu().numpy(), fused_series=fused[1].cpu().numpy())
            print(f"[gpu] 1000-step chain {k}: done", flush=True)
        for j in TAPS:
            record(f"1000-step DDPM chain, cfg 9, B={B}: x after loop index {j}", [g[f"ref_x{j}"]], [gpu["f32"]["taps"][j]],
                   [gpu["bf16x3"]["taps"][j]], [g[f"cpu32_x{j}"]])
        record(f"1000-step chain, B={B}: fused sampler final latent", [g["ref_x999"]], [gpu["f32"]["fused_latent"]],
               [gpu["bf16x3"]["fused_latent"]], [g["cpu32_x999"]])
        record(f"1000-step chain, B={B}: decoded series (B,96)", [g["ref_series"]], [gpu["f32"]["fused_series"]],
               [gpu["bf16x3"]["fused_series"]], [g["cpu32_series"]])

    n_ok = sum(r["x3_meets_bar"] for r in table)
    n_ok_f32 = sum(r["f32_meets_bar"] for r in table)
    failing = [r["case"] for r in table if not r["x3_meets_bar"]]
    out = {"reference": "fp64 arithmetic of oracle/t2s_oracle.py on the fp32 weights / inputs / schedule / time embedding",
           "bar": f"rms(bf16x3) <= {RMS_BAR} x rms(CPU fp32 oracle) and max(bf16x3) <= {MAX_BAR} x max(CPU fp32 oracle), every entry",
           "host_tables_this_box_vs_reference_box": host_diff, "rows": table, "entries": len(table), "bf16x3_entries_meeting_bar": n_ok, "f32_mfma_entries_meeting_bar": n_ok_f32,
           "bf16x3_meets_bar_everywhere": n_ok == len(table), "bf16x3_failing_entries": failing}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    path = os.path.join(REPO, "gpurun_out", f"{args.tag}_accuracy")
    json.dump(out, open(path + ".json", "w"), indent=1)
    lines = ["| case | elements | max abs of fp64 ref | CPU fp32 oracle rms / max | f32 MFMA rms / max | bf16x3 rms / max | bf16x3 / oracle rms, max | f32 MFMA / oracle rms, max | bar |",
             "|---|---|---|---|---|---|---|---|---|"]
    for r in table:
        lines.append(f"| {r['case']} | {r['bf16x3']['n']} | {r['bf16x3']['ref_max_abs']:.3g} | {r['cpu_fp32_oracle']['rms']:.3e} / "
                     f"{r['cpu_fp32_oracle']['max_abs']:.3e} | {r['f32_mfma']['rms']:.3e} / {r['f32_mfma']['max_abs']:.3e} | "
                     f"{r['bf16x3']['rms']:.3e} / {r['bf16x3']['max_abs']:.3e} | **{r['x3_over_oracle_rms']:.3f}**, **{r['x3_over_oracle_max']:.3f}** | "
                     f"{r['f32_over_oracle_rms']:.3f}, {r['f32_over_oracle_max']:.3f} | {'ok' if r['x3_meets_bar'] else 'FAIL'} |")
    lines.append("")
    lines.append(f"Bar: {out['bar']}.  bf16x3 meets it in {n_ok} of {len(table)} entries (f32 MFMA: {n_ok_f32} of {len(table)})."
                 + ("" if not failing else "  Failing: " + "; ".join(failing)))
    open(path + ".md", "w").write("\n".join(lines) + "\n")
    print("\n" + "\n".join(lines))
    print(f"wrote {path}.json / .md")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=("cpu", "gpu", "all"), default="gpu")
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--small", action="store_true", help="the round-3 sample: B = 8 / 4 / 2, one seed")
    ap.add_argument("--skip-1000", action="store_true")
    ap.add_argument("--force", action="store_true", help="stage cpu: recompute references that already exist")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    args = ap.parse_args()
    if args.stage in ("cpu", "all"):
        stage_cpu(args)
    if args.stage in ("gpu", "all"):
        stage_gpu(args)


if __name__ == "__main__":
    main()
