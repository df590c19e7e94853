#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4xchg
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=short -W always -k "split_program or grouped_program or headline_batch or small_batch or group_formation" 2>&1 | grep -i "first report\|FAILED\|passed\|failed\|Error" | cut -c1-300 | head -8
done
timeout 300 python tools/time_cfg2.py 256 32 8 128 192 2>&1 | grep -v amdgpu | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rs --tb=short -W always > gpurun_out/r4xchg/full.log 2>&1
grep -n "FAILED\|SKIPPED\|first report" gpurun_out/r4xchg/full.log | head -20; tail -2 gpurun_out/r4xchg/full.log
