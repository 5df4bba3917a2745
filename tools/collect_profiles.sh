#!/bin/bash
# Run on the GPU box (via gpurun): the headline bench, rocprofv3 kernel stats of the same command, the HBM-traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, per MI355X_MICROARCH.md), the MFMA-busy counters of the dominant
# kernel, and the same for the bf16 training step.   Usage: tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*
# Afterwards (in the build container): cp gpurun_out/<tag>_* profiles/ && python tools/summarize_profiles.py <tag>
set -o pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
mkdir -p $out
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
echo "== bench"; timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 1 > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
tail -c 400 $out/${tag}_bench.json; echo
# (--no-pmc: a profiled bench must not start profiler children of its own; --no-configs: the profile is of the headline shape)
S="--gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-math --no-train --no-strong --no-legs --no-configs --no-pmc"
echo "== rocprofv3 --kernel-trace --stats (sampling only, 1+1 batches; two sampler lanes = the default)"
rm -rf /tmp/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py $S > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err || exit 1
cp /tmp/prof_$tag/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
# one lane: every kernel alone on the chip at the 512-sequence launch shape -- the configuration the roofline block
# of bench.py is quoted on (per-kernel durations of the two-lane run include time-sharing with the other lane)
echo "== rocprofv3 --kernel-trace --stats, --lanes 1"
rm -rf /tmp/prof1_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$tag -- python3 bench.py $S --lanes 1 > $out/${tag}_prof_bench_lanes1.json 2> $out/${tag}_prof1.err || exit 1
cp /tmp/prof1_$tag/*/*kernel_stats.csv $out/${tag}_kernel_stats_lanes1.csv
head -8 $out/${tag}_kernel_stats_lanes1.csv

# the 32-series shard of an 8-GPU strong-scaling run (256 series / 8): launch shapes of its own (two lanes of 16 series:
# 16-token row chain, the two-workgroups-per-head attention kernel)
echo "== rocprofv3 --kernel-trace --stats, --batch 32 (strong-scaling shard)"
rm -rf /tmp/prof32_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof32_$tag -- python3 bench.py $S --no-strong --batch 32 > $out/${tag}_prof_bench_b32.json 2> $out/${tag}_prof32.err || exit 1
cp /tmp/prof32_$tag/*/*kernel_stats.csv $out/${tag}_kernel_stats_b32.csv
head -8 $out/${tag}_kernel_stats_b32.csv

agg_pmc() {   # <counter dir> <dst csv>: per kernel and counter, dispatches and the per-dispatch average
  python3 - "$1" "$2" <<'PY'
import csv, glob, sys, collections
d, dst = sys.argv[1:3]
f = sorted(glob.glob(d + '/*/*counter_collection.csv'))[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    a = agg[(r['Kernel_Name'], r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
with open(dst, 'w') as o:
    o.write('kernel,counter,dispatches,avg_per_dispatch\n')
    for (k, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        o.write('"%s",%s,%d,%.1f\n' % (k, c, n, v / n))
print(open(dst).read()[:600])
PY
}
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (sampling, one lane, 3 diffusion steps)"
  rm -rf /tmp/pmc_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py $S --diffusion-steps 3 --lanes 1 > /dev/null 2> $out/${tag}_pmc_$c.err || exit 1
  agg_pmc /tmp/pmc_$c $out/${tag}_pmc_$c.csv
done
echo "== pmc SQ busy counters (sampling, one lane, 3 diffusion steps)"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pmcs_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcs_$i -- python3 bench.py $S --diffusion-steps 3 --lanes 1 > /dev/null 2> $out/${tag}_pmc_sq$i.err || { echo "pass failed: $grp"; continue; }
  agg_pmc /tmp/pmcs_$i $out/${tag}_pmc_sq$i.csv
done

# ---- training step (BASELINE configs[3] shape, bf16, cached latents): the bench leg alone, its kernel stats, its HBM bytes
echo "== training bench"
timeout -k 10 300 python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $out/${tag}_train_bench.json || exit 1
cut -c1-400 $out/${tag}_train_bench.json
rm -rf /tmp/proft_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$tag -- python3 tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_train_prof.err || exit 1
cp /tmp/proft_$tag/*/*kernel_stats.csv $out/${tag}_train_bf16_kernel_stats.csv
head -12 $out/${tag}_train_bf16_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c (training: 1 warm-up + 3 timed + 3 event-timed steps)"
  rm -rf /tmp/pmct_$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmct_$c -- python3 tools/bench_train.py --steps 3 --warmup 1 > /dev/null 2> $out/${tag}_train_pmc_$c.err || exit 1
  agg_pmc /tmp/pmct_$c $out/${tag}_train_pmc_$c.csv
done

# ---- opt-in bf16x3 arithmetic (not the headline): kernel stats of a 50-step run
echo "== bf16x3 kernel stats"
rm -rf /tmp/profx_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profx_$tag -- python3 bench.py $S --math bf16x3 --diffusion-steps 50 > /dev/null 2> $out/${tag}_x3_prof.err || exit 1
cp /tmp/profx_$tag/*/*kernel_stats.csv $out/${tag}_x3_kernel_stats.csv
head -6 $out/${tag}_x3_kernel_stats.csv
echo "== done"
