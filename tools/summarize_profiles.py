"""Write profiles/<tag>_summary.md, profiles/attn_pmc.json and profiles/train_traffic.json from the files
tools/collect_profiles.sh <tag> produced (copied into profiles/).

    python tools/summarize_profiles.py r02_v1

The two json files are what bench.py reports as `roofline.traffic` / `roofline.pmc` (headline) and `train.roofline.traffic`:
counter values that need rocprofv3 passes of their own, so they are read from here and carry the tag they came from.
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports 2x on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import json
import os
import sys

tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md peak engine clock; profiles/r01_clocks_under_load.txt measured 2.39 under load
N_SIMD = 1024


def stats(name):
    return list(csv.DictReader(open(os.path.join(P, f"{tag}_{name}.csv"))))


def table(rows, top=10):
    out = ["| kernel | calls | avg us | % |", "|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| `%s` | %s | %.1f | %.1f |" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(out)


def pmc(name):
    """{(kernel, counter): (dispatches, avg)}"""
    path = os.path.join(P, f"{tag}_{name}.csv")
    if not os.path.exists(path):
        return {}
    return {(r["kernel"], r["counter"]): (int(r["dispatches"]), float(r["avg_per_dispatch"])) for r in csv.DictReader(open(path))}


def find(d, kernel_part, counter):
    for (k, c), v in d.items():
        if kernel_part in k and c == counter:
            return v
    return None


b = json.load(open(os.path.join(P, f"{tag}_bench.json")))
rf, cb, am, tr = b["roofline"], b.get("cpu_baseline"), b.get("alt_math"), b.get("train")
k2, k1 = stats("kernel_stats"), stats("kernel_stats_lanes1")
attn1 = next(float(r["AverageNs"]) / 1e3 for r in k1 if "attn_fwd_persistent" in r["Name"])
fs, ws = pmc("pmc_FETCH_SIZE"), pmc("pmc_WRITE_SIZE")
fa, wa = find(fs, "attn_fwd_persistent", "FETCH_SIZE"), find(ws, "attn_fwd_persistent", "WRITE_SIZE")
attn_bytes = (2 * fa[1] + wa[1]) * 1024 if fa and wa else None
sq = {}
sq.update(pmc("pmc_sq1"))
sq.update(pmc("pmc_sq2"))
busy = find(sq, "attn_fwd_persistent", "SQ_VALU_MFMA_BUSY_CYCLES")
mfma_busy = busy[1] / N_SIMD / (attn1 * 1e-6 * CLOCK_GHZ * 1e9) if busy else None
json.dump({"kernel": "attn_fwd_persistent_kernel", "hbm_bytes_per_launch": attn_bytes,
           "fetch_size_kb_raw": fa[1] if fa else None, "write_size_kb": wa[1] if wa else None,
           "mfma_busy_frac": mfma_busy,
           "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES per dispatch / 1024 SIMDs / (rocprofv3 average duration of the one-lane run x 2.4 GHz)",
           "avg_launch_us_rocprof_lanes1": attn1,
           "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv, {tag}_pmc_WRITE_SIZE.csv, {tag}_pmc_sq1.csv (separate rocprofv3 --pmc passes of "
                     f"`bench.py --lanes 1 --diffusion-steps 3`, tools/collect_profiles.sh {tag}); FETCH_SIZE x2 (gfx950 correction)"},
          open(os.path.join(P, "attn_pmc.json"), "w"), indent=1)

lines = [f"# {tag}: MI355X, collected by `tools/collect_profiles.sh {tag}` in one gpurun call", "",
         "## Headline: sampling (BASELINE configs[1]), f32 MFMA",
         f"`python bench.py --gpus 1 --steps 3 --warmup 1` (`profiles/{tag}_bench.json`): **{b['value']:.1f} series/s**, "
         f"{b['ms_per_step']:.0f} ms per 256-series batch, sampler lanes = {b['config'].get('sampler_lanes')}; whole path "
         f"{b['whole_path_tflops']:.1f} TFLOP/s = {b['whole_path_frac_of_fp32_mfma_peak']:.3f} of the fp32 MFMA peak."]
if cb:
    lines.append(f"CPU baseline (oracle, fused SDPA): {cb['value']:.4f} series/s on {cb['cores']} cores ({b['gpu_over_cpu']:.0f}x), "
                 f"{cb['one_thread']['value']:.4f} series/s on one thread.")
lines += ["",
          f"Roofline block (dominant kernel, alone on the chip at the 512-sequence launch shape): attention {rf['avg_launch_us']:.1f} us "
          f"in situ = {rf['achieved']:.1f} TFLOP/s = **{rf['frac']:.3f}** of peak; rocprofv3 average of the one-lane run below: {attn1:.1f} us "
          f"= {60397977600 / (attn1 * 1e-6) / 1e12 / 157.3:.3f}.",
          "",
          "### One lane (`--lanes 1`): every kernel alone on the chip -- the configuration the roofline block is quoted on",
          "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-train --lanes 1`",
          "", table(k1), "",
          "### Two lanes (the default, the timed region of the headline): same command without `--lanes 1`",
          "Each kernel is issued as two 256-sequence launches, one per lane, that time-share the CUs with the other lane's kernels: "
          "the per-launch averages below include that sharing and are not kernel properties; what the pipelining buys is the batch time above.",
          "", table(k2), ""]
if attn_bytes:
    lines += [f"HBM traffic of the dominant kernel (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes on the one-lane run): attention "
              f"2 x {fa[1]:,.0f} KB + {wa[1]:,.0f} KB = {attn_bytes / 1e6:.1f} MB per launch against 503.3 MB of algorithmic bytes "
              f"(q, k, v in, o out for 2048 heads).", ""]
if mfma_busy:
    lines += [f"Matrix-pipe busy share of the same kernel: SQ_VALU_MFMA_BUSY_CYCLES {busy[1]:,.0f} per dispatch = **{mfma_busy:.3f}** of the kernel's "
              f"SIMD-cycles (formula in `profiles/attn_pmc.json`).", ""]
rows_busy = [(k, v) for (k, c), v in sq.items() if c == "SQ_VALU_MFMA_BUSY_CYCLES" and "dit_rows" in k]
if rows_busy:
    lines += ["Row-chain kernels, same pass: " + "; ".join(f"`{k[:60]}` {v[1]:,.0f} busy cycles / dispatch" for k, v in rows_busy), ""]
rr = b.get("roofline_rows")
if rr:
    lines += ["Row chain against the fp32 MFMA peak (`roofline_rows`, same in-situ timing): " +
              "; ".join(f"`{k.split(' ')[0]}` {v['avg_launch_us']:.1f} us = {v['achieved']:.1f} TFLOP/s = **{v['frac']:.3f}**" for k, v in rr["instances"].items()) +
              f"; all five launches of a forward: {rr['achieved']:.1f} TFLOP/s = **{rr['frac']:.3f}**.", ""]
    if rr.get("pmc"):
        lines += ["Their HBM bytes per launch, measured by the bench run itself (child `rocprofv3 --pmc` passes): " +
                  "; ".join(f"`{k}` {v['traffic'] / 1e6:.0f} MB" for k, v in rr["pmc"]["instances"].items()) + ".", ""]
pm_ = (rf.get("pmc") or {})
if pm_.get("measured_by_this_run"):
    lines += [f"The bench run's OWN PMC passes (`roofline.pmc.measured_by_this_run`): attention {rf['traffic'] / 1e6:.1f} MB per launch, matrix pipe busy "
              f"{pm_['mfma_busy_frac']:.3f}.", ""]
for key, title in (("config3", "BASELINE configs[2] under the bench clock"), ("config5", "BASELINE configs[4] under the bench clock")):
    c = b.get(key)
    if c and "value" in c:
        lines += [f"## {title} (`{key}`)", f"{c['metric']}: **{c['value']:.1f} series/s**, whole path {c['whole_path_tflops']:.1f} TFLOP/s = "
                  f"{c['whole_path_frac_of_fp32_mfma_peak']:.3f} of the fp32 MFMA peak." +
                  ("  Per length: " + ", ".join(f"L={L_} {v['series_per_s']:.1f}" for L_, v in c["lengths"].items()) + " series/s." if "lengths" in c else ""), ""]
if am:
    lines += ["## bf16x3 arithmetic (include/t2s.h T2S_MATH_BF16X3; DESIGN.md 4.4): the drivers' default, not the bench headline",
              f"`bench.py` reports it as `alt_math`: **{am['value']:.1f} series/s** ({am['ms_per_step']:.0f} ms per batch); attention "
              f"{am['attention_us']:.0f} us alone on the chip, row chain {am['row_chain_us']:.0f} us average.  rocprofv3 kernel stats of "
              f"`bench.py --math bf16x3 --diffusion-steps 50` (`{tag}_x3_kernel_stats.csv`):", "", table(stats("x3_kernel_stats"), 6), ""]
    if am.get("roofline"):
        r3 = am["roofline"]
        lines += [f"Its own roofline (dense bf16 peak / 6 = {r3['peak']:.0f} algorithmic TFLOP/s: an fp32-accurate product is six bf16 MFMAs): attention "
                  f"{r3['achieved']:.1f} = **{r3['frac']:.3f}**, whole path {r3['whole_path_tflops']:.1f} = {r3['whole_path_frac']:.3f}.", ""]
    if am.get("accuracy_vs_fp64") and "bf16x3" in am["accuracy_vs_fp64"]:
        a = am["accuracy_vs_fp64"]
        lines += ["In-run accuracy block (one conditional forward, B = 32, against the fp64 arithmetic of the oracle), rms / max: " +
                  ", ".join(f"{k} {a[k]['rms']:.3e} / {a[k]['max_abs']:.3e}" for k in ("cpu_fp32_oracle", "f32_mfma", "bf16x3")) +
                  f"; bf16x3 / oracle = {a['bf16x3_over_oracle']['rms']:.3f} (rms), {a['bf16x3_over_oracle']['max_abs']:.3f} (max).", ""]
    if am.get("strong_shards"):
        s3 = am["strong_shards"]
        lines += ["Strong-scaling shards in this arithmetic (series/s on 128 / 64 / 32 series): " +
                  " / ".join(f"{s3['shards'][n]['series_per_s']:.1f}" for n in ("128", "64", "32")) +
                  "; predicted efficiency at 2 / 4 / 8 GPUs " + " / ".join(f"{s3['predicted_strong_efficiency'][w]:.3f}" for w in ("2", "4", "8")) +
                  " (no 16-token row kernel in this arithmetic: a bf16 MFMA's internal summation cannot be made bit-identical across tile shapes).", ""]

# ---- training
tb_path = os.path.join(P, f"{tag}_train_bench.json")
if os.path.exists(tb_path):
    t = json.load(open(tb_path))
    kt = stats("train_bf16_kernel_stats")
    tf, tw = pmc("train_pmc_FETCH_SIZE"), pmc("train_pmc_WRITE_SIZE")
    n_steps = 1 + 3 + 3          # tools/bench_train.py --steps 3 --warmup 1, plus the 3 event-timed steps of the class breakdown
    tot_f = sum(n * v for (k, c), (n, v) in tf.items() if c == "FETCH_SIZE")
    tot_w = sum(n * v for (k, c), (n, v) in tw.items() if c == "WRITE_SIZE")
    per_step = (2 * tot_f + tot_w) * 1024 / n_steps if tf and tw else None
    json.dump({"per_gpu_batch": t["per_gpu_batch"], "hbm_bytes_per_step": per_step,
               "source": f"profiles/{tag}_train_pmc_FETCH_SIZE.csv + _WRITE_SIZE.csv: all kernels of `tools/bench_train.py --steps 3 --warmup 1` "
                         f"({n_steps} training steps incl. the event-timed ones; the one-off latent-cache fill and the few torch kernels are "
                         f"included, < 1 %); FETCH_SIZE x2 (gfx950 correction)"},
              open(os.path.join(P, "train_traffic.json"), "w"), indent=1)
    lines += ["## Training step (BASELINE configs[3] shape: bf16, B=1152/GPU, L=96, DDPM T=100, cached latents)",
              f"`tools/bench_train.py --steps 20 --warmup 3` = the `train` leg of bench.py (`{tag}_train_bench.json`): **{t['ms_per_step']:.2f} ms/step, "
              f"{t['value']:,.0f} samples/s**, {t['tflops_algorithmic']:.0f} algorithmic TFLOP/s.",
              "", "Per class, HIP events in situ (ms per step): " + ", ".join(f"{k} {v['ms_per_step']:.2f}" for k, v in t["kernel_classes"].items()), ""]
    if per_step:
        lines += [f"HBM bytes per step from the PMC passes: **{per_step / 1e9:.1f} GB** (design count {t['hbm_bytes_per_step_model'] / 1e9:.1f} GB) "
                  f"= {per_step / (t['ms_per_step'] * 1e-3) / 1e12:.2f} TB/s over the whole step.", ""]
    lines += [f"rocprofv3 kernel stats (`{tag}_train_bf16_kernel_stats.csv`; 1 warm-up + 3 timed + 3 event-timed steps):", "", table(kt, 26), ""]
    # per-kernel HBM roofline: bytes by design per launch (units u of bench.py's TRAIN_U: one bf16 (M,128) tensor) / average duration
    u_mb = t["per_gpu_batch"] * 480 * 128 * 2 / 1e6
    per_launch_u = [("bgemm_kernel<128, 384, 1, 1>", 6, "qkv GEMM + LN prologue, block 0"),
                    ("bgemm_kernel<128, 384, 3, 1>", 9, "qkv GEMM, LN prologue + previous gate/residual add"),
                    ("attn16_fwd_kernel", 4, "attention forward (VALU-bound)"),
                    ("bgemm_kernel<128, 128, 0, 0>", 2, "proj GEMM / proj data gradient"),
                    ("bgemm_kernel<128, 256, 3, 0>", 8, "fc1 GEMM, LN prologue + attention gate/residual add"),
                    ("bgemm_kernel<256, 128, 2, 0>", 3, "fc2 GEMM, GELU prologue"),
                    ("wgrad16_kernel<true>", 4, "fc2 weight gradient (df read twice)"),
                    ("bgemm_kernel<128, 256, 0, 2>", 5, "fc2 data gradient x gelu'"),
                    ("wgrad16_kernel<false>", 11.0 / 3, "fc1 / proj / qkv weight gradients (average; X of fc1 read twice, of qkv three times)"),
                    ("bgemm_kernel<256, 128, 0, 3>", 10, "fc1 data gradient + LN2 backward + gate backward"),
                    ("bgemm_kernel<384, 128, 0, 3>", 10.5, "qkv data gradient + LN1 backward + gate backward (block 0: 9 u)"),
                    ("attn16_bwd_dq_kernel", 6, "attention dQ + D_i (VALU-bound)"),
                    ("attn16_bwd_dkv_kernel", 6, "attention dK, dV (VALU-bound)")]
    lines += ["Per kernel against the HBM roof (bytes by design per launch, `TRAIN_U` in bench.py, u = %.1f MB; 8 TB/s peak, ~6 TB/s is what a plain copy reaches):" % u_mb,
              "", "| kernel | what | u per launch | avg us | TB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
    for pat, uu, what in per_launch_u:
        r = next((r for r in kt if pat in r["Name"]), None)
        if r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        tbs = uu * u_mb * 1e6 / (us * 1e-6) / 1e12
        lines.append("| `%s` | %s | %.1f | %.1f | %.2f | %.2f |" % (pat, what, uu, us, tbs, tbs / 8.0))
    lines.append("")
# ---- strong-scaling shards (BASELINE's metric read literally: 256 series at 1/2/4/8 GPUs)
ss = b.get("strong_shards")
if ss:
    lines += ["## Strong-scaling shards: the per-GPU share of a 256-series job, timed on this one GPU (`strong_shards` in the bench line)",
              "", "| GPUs | series per GPU | series/s per GPU | ms per batch | predicted efficiency | predicted job rate (series/s) |",
              "|---|---|---|---|---|---|",
              f"| 1 | {b['config']['global_batch']} | {b['value']:.2f} | {b['ms_per_step']:.0f} | 1.000 | {b['value']:.1f} |"]
    for n, r in sorted(ss["shards"].items(), key=lambda kv: -int(kv[0])):
        w = str(r["gpus_for_256_total"])
        lines.append(f"| {w} | {n} | {r['series_per_s']:.2f} | {r['ms_per_batch']:.0f} | {ss['predicted_strong_efficiency'][w]:.3f} | "
                     f"{ss['predicted_strong_series_per_s'][w]:.1f} |")
    lines += ["", "Sampling has no data-path collective (rows are independent, Philox keyed by the global row): N x rate(256 / N) is the N-GPU "
              "rate up to launch skew between ranks.", ""]
b32 = os.path.join(P, f"{tag}_kernel_stats_b32.csv")
if os.path.exists(b32):
    lines += ["### The 32-series shard under rocprofv3 (`bench.py --batch 32`: two lanes of 16 series, 16-token row chain, two-workgroups-per-head attention)",
              "", table(stats("kernel_stats_b32"), 9), ""]
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:16]))
