#!/bin/bash
# round 6, call B: streaming weight ring (CDX2_STREAM_RING) -- parity subset, timings, op profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or test_fused_sample_matches_reference_fixture or group or split" 2>&1 | tail -3
timeout 300 python tools/time_cfg2.py 256 32 512 3200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6b/time_cfg2.txt
timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6b/op_profile_group4.txt
tail -3 gpurun_out/r6b/op_profile_group4.txt
