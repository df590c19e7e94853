#!/bin/bash
# round 5, call D: are seeded CPU draws host-independent (config-4 yardstick question), the new training nodes (DiT1d / IDQLMlp / ChiUNet1d),
# the fault hook, and a fresh op profile of the grouped program.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
python tools/randn_host_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5d/randn_host_check.txt
grep -m1 "model name" /proc/cpuinfo | tee -a gpurun_out/r5d/randn_host_check.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s -k "training or lost_granule or layernorm_and_attention or loss_and_update or device_query or update_runs" 2>&1 > gpurun_out/r5d/gpu_subset.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5d/gpu_subset.txt | head -40
grep -B2 -A30 "^___" gpurun_out/r5d/gpu_subset.txt | head -200 > gpurun_out/r5d/gpu_subset_failures.txt
timeout 300 python tools/op_profile2.py 256 group4 > gpurun_out/r5d/op_profile_group4.txt 2>&1; tail -3 gpurun_out/r5d/op_profile_group4.txt
