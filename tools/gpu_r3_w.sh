#!/bin/bash
mkdir -p gpurun_out/r3w
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "empty_and_ragged or training_step or beyond_one_workgroup or shipped_large or guided" 2>&1 | tail -80 > gpurun_out/r3w/tests.log
cat gpurun_out/r3w/tests.log
