#!/bin/bash
# Shader clock and socket power while the attention kernels run back to back (power limiting?).
# usage: tools/probe_clocks.sh   (on the GPU box)
python3 - <<'PY' &
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from t2ms_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
dev = torch.device("cuda:0")
BH = 2048
for kind, mk in (("randn", lambda: torch.randn(BH, 480, 32, device=dev)), ("zeros", lambda: torch.zeros(BH, 480, 32, device=dev))):
    q, k, v = mk(), mk(), mk()
    o = torch.empty_like(q)
    p = [C.c_void_p(t.data_ptr()) for t in (q, k, v, o)]
    for name, fn, n in (("f32", lambda: lib.t2s_attn_fwd_packed(*p, BH // 4, None), 6000), ("x3", lambda: lib.t2s_attn_fwd_x3(*p, BH, None), 6000)):
        t0 = time.time()
        print("PHASE", kind, name, "start", flush=True)
        i = 0
        while time.time() - t0 < 4.0:
            for _ in range(50): fn()
            torch.cuda.synchronize(); i += 50
        print("PHASE", kind, name, "end calls", i, "avg us %.1f" % ((time.time() - t0) / i * 1e6), flush=True)
PY
pid=$!
sleep 8   # torch import
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | tr '\n' ' '; echo
  sleep 0.5
  kill -0 $pid 2>/dev/null || break
done
wait $pid
