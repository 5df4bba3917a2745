for a in 0 1 2; do
  out=$(T2S_ATTN_FWD_PC=1 T2S_FB_ABL=$a python tools/bench_train.py --steps 10 --warmup 2 2>/dev/null)
  echo "$out" | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_classes']
print('ABL=$a'.ljust(10), round(t['ms_per_step'],3), 'ms', {a[6:]: round(v['ms_per_step'],2) for a,v in k.items()})"
done
