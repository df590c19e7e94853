#!/bin/bash
mkdir -p gpurun_out/r3x
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "conditional_request_with_classifier or binding_stub" 2>&1 | tail -40 > gpurun_out/r3x/tests.log
cat gpurun_out/r3x/tests.log
