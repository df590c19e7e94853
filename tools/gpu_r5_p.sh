#!/bin/bash
# round 5, call P: parameter gradients accumulated in place inside update()
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "update or training or train or graph or halfdit or optim or adamw" 2>&1 | tail -6 | tee gpurun_out/r5p/tests.txt
timeout 600 python tools/update_bench.py 2>&1 | grep "update()" | tee gpurun_out/r5p/update_bench.txt
timeout 200 python tools/update_census.py cfg2 2>&1 | grep -v Warning > gpurun_out/r5p/census_cfg2.txt; head -12 gpurun_out/r5p/census_cfg2.txt
