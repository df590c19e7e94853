#!/bin/bash
# round 6, call C: host-side A/B through environment switches (same library): ENVS="A=1 B=0" ...; config 2 at B = 256, then the op profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 400 python tools/time_cfg2.py $SPECS 2>&1 | grep "traj/s" | cut -c1-170 | tee gpurun_out/r6c/time_${TAG:-x}.txt
if [ -n "$PROFILE" ]; then
  timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6c/op_profile_group4_${TAG:-x}.txt; tail -1 gpurun_out/r6c/op_profile_group4_${TAG:-x}.txt
fi
if [ -n "$CHECK" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "unet2 or three_traj or test_fused_sample_matches_reference_fixture or group or split" 2>&1 | tail -3
fi
