#!/bin/bash
# Run on the GPU box (via gpurun): headline bench, rocprofv3 kernel stats of the same command, and
# the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, per MI355X_MICROARCH.md).
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
set -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
mkdir -p $out
echo "== bench"; timeout -k 10 500 python bench.py --gpus 1 --steps 3 --warmup 1 > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
tail -c 600 $out/${tag}_bench.json; echo
echo "== rocprofv3 --kernel-trace --stats (same command, 1+1 batches; two sampler lanes = the default)"
rm -rf /tmp/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err || exit 1
cp /tmp/prof_$tag/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
# one lane: every kernel alone on the chip at the 512-sequence launch shape -- the configuration the roofline block
# of bench.py is quoted on (per-kernel durations of the two-lane run include time-sharing with the other lane)
echo "== rocprofv3 --kernel-trace --stats, --lanes 1"
rm -rf /tmp/prof1_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$tag -- python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --lanes 1 > $out/${tag}_prof_bench_lanes1.json 2> $out/${tag}_prof1.err || exit 1
cp /tmp/prof1_$tag/*/*kernel_stats.csv $out/${tag}_kernel_stats_lanes1.csv
head -8 $out/${tag}_kernel_stats_lanes1.csv
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rm -rf /tmp/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python bench.py --gpus 1 --steps 1 --warmup 1 --diffusion-steps 3 --no-cpu-baseline --no-alt-math --lanes 1 > /dev/null 2> $out/${tag}_pmc_$c.err || exit 1
  python3 - "$c" /tmp/pmc_$c $out/${tag}_pmc_$c.csv <<'PY'
import csv, glob, sys, collections
c, d, dst = sys.argv[1:4]
f = sorted(glob.glob(d + '/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == c:
        a = agg[r['Kernel_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
with open(dst, 'w') as o:
    o.write('kernel,counter,dispatches,avg_per_dispatch\n')
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        o.write('"%s",%s,%d,%.1f\n' % (k, c, n, v / n))
print(open(dst).read()[:700])
PY
done

# ---- training step (BASELINE config 4 shape): both arithmetic modes, with and without the latent cache,
# and the rocprofv3 kernel stats of the bf16 / cached step
echo "== training bench"
: > $out/${tag}_train_bench.jsonl
for mode in "f32 --cache_latents" "bf16" "bf16 --cache_latents"; do
  timeout -k 10 300 python tools/bench_train.py --batch 1152 --steps 5 --warmup 2 --dtype $mode 2>/dev/null | tail -1 >> $out/${tag}_train_bench.jsonl || exit 1
done
cat $out/${tag}_train_bench.jsonl
rm -rf /tmp/proft_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$tag -- python3 tools/bench_train.py --batch 1152 --steps 3 --warmup 1 --dtype bf16 --cache_latents > /dev/null 2> $out/${tag}_train_prof.err || exit 1
cp /tmp/proft_$tag/*/*kernel_stats.csv $out/${tag}_train_bf16_kernel_stats.csv
head -12 $out/${tag}_train_bf16_kernel_stats.csv

# ---- opt-in bf16x3 arithmetic (not the headline): kernel stats of a 50-step run
echo "== bf16x3 kernel stats"
rm -rf /tmp/profx_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profx_$tag -- python3 bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --math bf16x3 --diffusion-steps 50 > /dev/null 2> $out/${tag}_x3_prof.err || exit 1
cp /tmp/profx_$tag/*/*kernel_stats.csv $out/${tag}_x3_kernel_stats.csv
head -6 $out/${tag}_x3_kernel_stats.csv
