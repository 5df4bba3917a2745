for b in 32 64; do for l in 1 2 4; do
python bench.py --math bf16x3 --batch $b --lanes $l --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); print('x3 batch $b lanes $l', round(t['value'],2), 'series/s')"
done; done
