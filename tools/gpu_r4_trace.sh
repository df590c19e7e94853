#!/bin/bash
cd $GRAFT_REPO_ROOT
for shape in "65536 960 320" "65536 1280 320 gelu_tanh" "65536 256 1280" "65536 256 320" "65536 1024 1024"; do
  timeout 120 python tools/gemm_trace.py $shape 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4trace.txt
