#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4xchg
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=short -W always -k "split_program_matches" 2>&1 | grep -v "^$" | grep -i "warn\|first report\|FAILED\|passed\|failed" | cut -c1-400 | head -12
done
