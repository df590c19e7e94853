#!/bin/bash
mkdir -p gpurun_out/r4f
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -q -p no:cacheprovider -m gpu > gpurun_out/r4f/pytest.log 2>&1
tail -25 gpurun_out/r4f/pytest.log
timeout 300 python tools/time_cfg2.py 256 256:CDX_UNET2_GROUP_MIN_KB=600 256:CDX_UNET2_GROUP_MIN_KB=700 256:CDX_UNET2_GROUP_MIN_KB=300 256:CDX_UNET2_GROUP=0 32:CDX_UNET2_SPLIT_SYNC=0 32 > gpurun_out/r4f/time.txt 2>&1
cat gpurun_out/r4f/time.txt
timeout 200 python tools/op_profile2.py 256 group4 > gpurun_out/r4f/op_profile_group4.txt 2>&1
tail -4 gpurun_out/r4f/op_profile_group4.txt
timeout 200 python tools/host_profile.py 32 > gpurun_out/r4f/host_profile.txt 2>&1
head -60 gpurun_out/r4f/host_profile.txt
