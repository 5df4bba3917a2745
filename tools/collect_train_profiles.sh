cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out; tag=r01_v7
: > $out/${tag}_train_bench.jsonl
for mode in "f32 --cache_latents" "bf16" "bf16 --cache_latents"; do
  timeout -k 10 300 python tools/bench_train.py --batch 1152 --steps 5 --warmup 2 --dtype $mode 2>/dev/null | tail -1 >> $out/${tag}_train_bench.jsonl || exit 1
done
cut -c1-190 $out/${tag}_train_bench.jsonl
rm -rf /tmp/proft_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$tag -- python3 tools/bench_train.py --batch 1152 --steps 3 --warmup 1 --dtype bf16 --cache_latents > /dev/null 2> $out/${tag}_train_prof.err || exit 1
cp /tmp/proft_$tag/*/*kernel_stats.csv $out/${tag}_train_bf16_kernel_stats.csv
