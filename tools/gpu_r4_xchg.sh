#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4xchg
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=short -k "split_program or grouped_program or headline_batch or small_batch" > gpurun_out/r4xchg/tests.log 2>&1
grep -n "FAILED\|SKIPPED\|Error" gpurun_out/r4xchg/tests.log | head -20; tail -3 gpurun_out/r4xchg/tests.log
for t in 0 256 512; do
  echo "CDX_UNET2_TUNE=$t"
  CDX_UNET2_TUNE=$t timeout 300 python tools/time_cfg2.py 256 32 128 2>&1 | grep -v amdgpu | cut -c1-200
done
