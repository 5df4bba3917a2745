"""Write profiles/<tag>_summary.md from the files tools/collect_profiles.sh <tag> produced (copied into profiles/).
    python tools/summarize_profiles.py r01_v7"""
import csv
import json
import os
import sys

tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def table(name, top=10):
    rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_{name}.csv"))))[:top]
    out = ["| kernel | calls | avg us | % |", "|---|---|---|---|"]
    for r in rows:
        out.append("| `%s` | %s | %.1f | %.1f |" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(out), {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}


def pmc(name):
    rows = list(csv.DictReader(open(os.path.join(P, f"{tag}_pmc_{name}.csv"))))
    return {r["kernel"]: float(r["avg_per_dispatch"]) for r in rows}


b = json.load(open(os.path.join(P, f"{tag}_bench.json")))
rf, cb, am = b["roofline"], b["cpu_baseline"], b.get("alt_math")
t2, _ = table("kernel_stats")
t1, k1 = table("kernel_stats_lanes1")
attn1 = next(v for k, v in k1.items() if "attn_fwd_persistent" in k)
fs, ws = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
fa = next(v for k, v in fs.items() if "attn_fwd_persistent" in k)
wa = next(v for k, v in ws.items() if "attn_fwd_persistent" in k)
lines = [f"# {tag}: MI355X, collected by `tools/collect_profiles.sh {tag}` in one gpurun call", "",
         "## Headline: sampling (BASELINE configs[1]), f32 MFMA",
         f"`python bench.py --gpus 1 --steps 3 --warmup 1` (`profiles/{tag}_bench.json`): **{b['value']:.1f} series/s**, "
         f"{b['ms_per_step']:.0f} ms per 256-series batch, sampler lanes = {b['config'].get('sampler_lanes')}; whole path "
         f"{b['whole_path_tflops']:.1f} TFLOP/s = {b['whole_path_frac_of_fp32_mfma_peak']:.3f} of the fp32 MFMA peak; CPU oracle "
         f"{cb['value']:.4f} series/s on {cb['cores']} cores ({b['gpu_over_cpu']:.0f}x).",
         "",
         f"Roofline block (dominant kernel, alone on the chip at the 512-sequence launch shape): attention {rf['avg_launch_us']:.1f} us "
         f"in situ = {rf['achieved']:.1f} TFLOP/s = **{rf['frac']:.3f}** of peak; rocprofv3 average of the one-lane run below: {attn1:.1f} us.",
         "",
         "### One lane (`--lanes 1`): every kernel alone on the chip -- the configuration the roofline block is quoted on",
         "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --lanes 1`",
         "", t1, "",
         "### Two lanes (the default, the timed region of the headline): same command without `--lanes 1`",
         "Each kernel is issued as two 256-sequence launches, one per lane, that time-share the CUs with the other lane's kernels: "
         "the per-launch averages below include that sharing (and queueing behind the other lane for the small kernels) and are "
         "not kernel properties; what the pipelining buys is the batch time above.", "", t2, "",
         f"HBM traffic of the dominant kernel (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes on the one-lane run, "
         f"`{tag}_pmc_*.csv`; FETCH_SIZE is in KB and under-reports 2x on gfx950): attention 2 x {fa:,.0f} KB + {wa:,.0f} KB = "
         f"{(2 * fa + wa) * 1024 / 1e6:.1f} MB per launch = its algorithmic bytes (q, k, v in, o out for 2048 heads) -> no re-reads.", ""]
if am:
    lines += ["## Opt-in bf16x3 arithmetic (include/t2s.h T2S_MATH_BF16X3; DESIGN.md 4.4) -- not the headline",
              f"`bench.py` reports it as `alt_math`: **{am['value']:.1f} series/s** ({am['ms_per_step']:.0f} ms per batch); attention "
              f"{am['attention_us']:.0f} us alone on the chip, row chain {am['row_chain_us']:.0f} us average.  rocprofv3 kernel stats of "
              f"`bench.py --math bf16x3 --diffusion-steps 50` (two lanes; `{tag}_x3_kernel_stats.csv`):", "", table("x3_kernel_stats", 6)[0], ""]
tb = [json.loads(l) for l in open(os.path.join(P, f"{tag}_train_bench.jsonl")) if l.strip()]
lines += ["## Training step (BASELINE config 4 shape, B=1152/GPU, L=96) -- `tools/bench_train.py`",
          "| dtype | latents | ms/step | samples/s |", "|---|---|---|---|"]
lines += ["| %s | %s | %.2f | %.0f |" % (t["dtype"], t["latents"], t["ms_per_step"], t["value"]) for t in tb]
lines += ["", f"rocprofv3 kernel stats of the bf16 / cached-latent step (`{tag}_train_bf16_kernel_stats.csv`; 1 warm-up + 3 timed steps; the "
          "single `vae_encode_kernel` call is the one-off latent-cache fill):", "", table("train_bf16_kernel_stats", 24)[0], ""]
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines[:14]))
