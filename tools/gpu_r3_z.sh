#!/bin/bash
mkdir -p gpurun_out/r3z
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "chitf_ta10" 2>&1 | grep -v "^  \|Warning" | tail -40 > gpurun_out/r3z/t.log
CDX_LN_VEC=0 timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "chitf_ta10" 2>&1 | tail -3 >> gpurun_out/r3z/t.log
cat gpurun_out/r3z/t.log
