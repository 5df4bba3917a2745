#!/bin/bash
# Where does a wave of the bf16x3 row-chain kernel spend its cycles?  tools/bin/libt2s_x3_stamp.so = the in-tree library with
# -DT2S_X3_STAMP (csrc/t2s_rows_x3.h: s_memtime stamps per category, first 256 workgroups, dumped once per kernel instance to stderr).
#   tools/x3_stamp.sh build     # build container (hipcc cross-compiles)
#   tools/x3_stamp.sh run       # GPU box: 60 eager 512-sequence CFG forwards in bf16x3
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/bin
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DT2S_X3_STAMP \
      -c t2ms_amd/csrc/t2s_dit.hip -o tools/bin/t2s_dit_x3_stamp.o
  objs=$(ls t2ms_amd/csrc/*.o | grep -v t2s_dit.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libt2s_x3_stamp.so $objs tools/bin/t2s_dit_x3_stamp.o
  ls -la tools/bin/libt2s_x3_stamp.so
else
  T2S_LIB=${T2S_LIB:-tools/bin/libt2s_x3_stamp.so} python - <<'PY'
import torch, bench
from t2ms_amd import _lib as L, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, _ = bench.build_models(dev)
model.set_math("bf16x3")
B = 256
x, text = synth.make_latents(3, B).to(dev), synth.make_text_embeddings(2025, B).to(dev)
lib = L.lib()
h = model.t2s_handle(dev, 2 * B)
st = torch.cuda.current_stream(dev).cuda_stream
temb = model.time_emb(torch.full((1,), 500, device=dev))
ou, oc = torch.empty_like(x), torch.empty_like(x)
for _ in range(60):
    L.check(lib.t2s_dit_forward_cfg(h, x.data_ptr(), temb.data_ptr(), text.data_ptr(), ou.data_ptr(), oc.data_ptr(), B, st))
torch.cuda.synchronize()
PY
fi
