#!/bin/bash
# round 4, call b: grouped mode first contact -- parity tests, A/B timing, op profile, DiT error budget
mkdir -p gpurun_out/r4e
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "grouped or headline_batch or split_program or beyond_one_tile or float64_yardstick or skips_parameters or steady_state" > gpurun_out/r4e/pytest.log 2>&1
tail -15 gpurun_out/r4e/pytest.log
timeout 300 python tools/time_cfg2.py 256 256:CDX_UNET2_GROUP=0 256:CDX_UNET2_GROUP=2 192 192:CDX_UNET2_GROUP=0 > gpurun_out/r4e/time.txt 2>&1
cat gpurun_out/r4e/time.txt
timeout 200 python tools/op_profile2.py 256 group4 > gpurun_out/r4e/op_profile_group4.txt 2>&1
tail -45 gpurun_out/r4e/op_profile_group4.txt
timeout 200 python tools/time_cfg2.py 32 64 128 > gpurun_out/r4e/time_small.txt 2>&1; cat gpurun_out/r4e/time_small.txt

