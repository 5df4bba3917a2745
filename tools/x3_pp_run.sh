set -o pipefail
export T2S_LIB=tools/bin/libt2s_x3_pp.so
T2S_X3_NW=8 timeout -k 10 180 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "x3" 2>&1 | tail -3 || exit 1
for r in 1 2 3; do
  for nw in 4 8; do
    T2S_X3_NW=$nw timeout -k 10 120 python bench.py --math bf16x3 --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_breakdown_us']; print('pipe lib, NW=$nw'.ljust(24), round(t['value'],2), 'series/s | attention', round(k['attention_x4'],1), 'rows avg', round(k['row_chain_x5'],1), 'us')" || exit 1
  done
done
unset T2S_LIB
timeout -k 10 120 python bench.py --math bf16x3 --steps 2 --warmup 1 --no-train --no-legs --no-strong --no-alt-math --no-cpu-baseline --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys; t=json.loads(sys.stdin.read()); k=t['kernel_breakdown_us']; print('in-tree (no pipe, NW=4)'.ljust(24), round(t['value'],2), 'series/s | rows avg', round(k['row_chain_x5'],1))"
T2S_X3_NW=8 T2S_LIB=tools/bin/libt2s_x3_pp_stamp.so timeout -k 10 120 bash tools/x3_stamp.sh run 2>&1 | grep x3_stamp
