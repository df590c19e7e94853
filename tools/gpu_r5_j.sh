#!/bin/bash
# round 5, call J: executor-independent seeded draws; whole GPU suite once more on the final tree.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short 2>&1 > gpurun_out/r5j/gpu_suite.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r5j/gpu_suite.txt | head -40
grep -B2 -A40 "^___" gpurun_out/r5j/gpu_suite.txt | head -250 > gpurun_out/r5j/gpu_suite_failures.txt
