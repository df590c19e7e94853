#!/bin/bash
mkdir -p gpurun_out/r4k
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "grouped or headline_batch or split_program or full_size_properties or steady_state" > gpurun_out/r4k/pytest.log 2>&1
tail -4 gpurun_out/r4k/pytest.log
timeout 300 python tools/time_cfg2.py 256 256:CDX_UNET2_GROUP=2 256:CDX_UNET2_GROUP=0 192 32 128 > gpurun_out/r4k/time.txt 2>&1
cat gpurun_out/r4k/time.txt
timeout 200 python tools/op_profile2.py 256 group4 > gpurun_out/r4k/op_profile_group4.txt 2>&1
tail -3 gpurun_out/r4k/op_profile_group4.txt
