#!/bin/bash
# round 5, call O: update() of the dp_pusht transformer (new native nodes) + kernel census of the config-2 step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
timeout 300 python tools/update_bench.py chitf 2>&1 | tail -1 | tee gpurun_out/r5o/update_chitf.txt
timeout 200 python tools/update_census.py cfg2 2>&1 | tail -40 | tee gpurun_out/r5o/census_cfg2.txt
timeout 200 python tools/update_census.py chitf 2>&1 | tail -30 | tee gpurun_out/r5o/census_chitf.txt
timeout 300 python -m pytest tests/test_dataset_siblings.py -m gpu -q 2>&1 | tail -2 | tee gpurun_out/r5o/dataset_tests.txt
