#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4diag
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rs --tb=short > gpurun_out/r4diag/full.log 2>&1
grep -n "FAILED\|SKIPPED\|Error\|assert" gpurun_out/r4diag/full.log | head -40
tail -5 gpurun_out/r4diag/full.log
timeout 300 python tools/time_cfg2.py 32 8 64 128 256 2>&1 | grep -v amdgpu | cut -c1-220
