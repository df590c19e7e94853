#!/bin/bash
# round 6, call A: baseline of the round on this box -- config-2 timings (B = 256 / 32 / 3200) and the op profile of the grouped program
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 300 python tools/time_cfg2.py 256 32 512 3200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6a/time_cfg2.txt
timeout 200 python tools/op_profile2.py 256 group4 2>&1 | grep -v amdgpu.ids > gpurun_out/r6a/op_profile_group4.txt
tail -3 gpurun_out/r6a/op_profile_group4.txt
