#!/bin/bash
# Per-segment cycle shares of the bf16x3 attention loop: builds the library with -DT2S_EXP=512 (cycle stamps),
# runs tools/probe_x3.py, rebuilds the normal library.  Run on the GPU box.
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
make -C t2ms_amd/csrc clean > /dev/null
make -C t2ms_amd/csrc FLAGS="$F -DT2S_EXP=512" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
timeout -k 10 120 python tools/probe_x3.py
rc=$?
make -C t2ms_amd/csrc clean > /dev/null
make -C t2ms_amd/csrc > /dev/null 2>&1
exit $rc
