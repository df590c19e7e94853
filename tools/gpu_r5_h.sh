#!/bin/bash
# round 5, call H: HalfDiT1d classifier gradient, conditional attention forwards, sibling datasets on the device.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "halfdit or attention or sibling or multi_horizon or dataset or resident" 2>&1 > gpurun_out/r5h/gpu_subset.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5h/gpu_subset.txt | head -40
grep -B2 -A40 "^___" gpurun_out/r5h/gpu_subset.txt | head -250 > gpurun_out/r5h/gpu_subset_failures.txt
