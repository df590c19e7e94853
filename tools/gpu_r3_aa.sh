#!/bin/bash
mkdir -p gpurun_out/r3aa
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "mlp or pearce or dql or sfbc or dvinv or baseline_cfg1 or empty_and_ragged" 2>&1 | tail -5 > gpurun_out/r3aa/tests.log
cat gpurun_out/r3aa/tests.log
timeout 300 python tools/bench_configs.py cfg1 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee gpurun_out/r3aa/cfg1.txt
