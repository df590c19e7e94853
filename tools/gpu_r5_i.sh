#!/bin/bash
# round 5, call I: the final bench line (live PMC passes pick the kernel by counter volume), multi-horizon dataset test on the device.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1200 python bench.py > gpurun_out/r5i/r05_bench_n1.json 2> gpurun_out/r5i/r05_bench_n1.err; head -c 1500 gpurun_out/r5i/r05_bench_n1.json; echo; tail -2 gpurun_out/r5i/r05_bench_n1.err
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "multi_horizon or sibling" 2>&1 | tail -3
