#!/bin/bash
# round 5, call E: whole GPU suite (same-box float64 yardsticks, DiT1d / IDQLMlp / ChiUNet1d training nodes), update() of configs 2-5 against
# ATen autograd, smoke.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -s 2>&1 > gpurun_out/r5e/gpu_suite_full.txt
grep -E "native vs fp64|beyond the elementwise|^(FAILED|ERROR)|passed|failed" gpurun_out/r5e/gpu_suite_full.txt | cut -c1-600 | head -40
grep -B2 -A30 "^___" gpurun_out/r5e/gpu_suite_full.txt | head -200 > gpurun_out/r5e/gpu_suite_failures.txt
timeout 600 python tools/update_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/update_bench.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5e/smoke.txt
