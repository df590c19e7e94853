#!/bin/bash
mkdir -p gpurun_out/r4o
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "dql_backprop or weighted_regression or loss_and_update" > gpurun_out/r4o/pytest.log 2>&1
tail -25 gpurun_out/r4o/pytest.log
