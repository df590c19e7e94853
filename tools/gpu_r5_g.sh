#!/bin/bash
# round 5, call G: training subset after the split-K scratch / census fix, update() of configs 2-5 (graph default, with / without split-K
# scratch, eager, ATen).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "update or training or graph or loss_and" 2>&1 > gpurun_out/r5g/gpu_subset.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5g/gpu_subset.txt | head -40
grep -B2 -A40 "^___" gpurun_out/r5g/gpu_subset.txt | head -250 > gpurun_out/r5g/gpu_subset_failures.txt
timeout 600 python tools/update_bench.py 2>&1 | grep -v "amdgpu.ids\|Synchronization debug\|_cuda_set_sync" | tee gpurun_out/r5g/update_bench.txt
