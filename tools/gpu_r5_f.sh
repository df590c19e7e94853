#!/bin/bash
# round 5, call F: HIP-graph update() by default (capturability probe) -- the training tests, update() of configs 2-5 three ways.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "update or training or graph or loss_and" 2>&1 > gpurun_out/r5f/gpu_subset.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5f/gpu_subset.txt | head -40
grep -B2 -A40 "^___" gpurun_out/r5f/gpu_subset.txt | head -250 > gpurun_out/r5f/gpu_subset_failures.txt
timeout 600 python tools/update_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f/update_bench.txt
