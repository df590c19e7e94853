#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4xchg
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rs --tb=short -W always > gpurun_out/r4xchg/full.log 2>&1
grep -n "FAILED\|SKIPPED\|first report" gpurun_out/r4xchg/full.log | head -20; tail -2 gpurun_out/r4xchg/full.log
for i in 1 2 3 4 5; do
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=short -W always -k "split_program or grouped_program or headline_batch or small_batch" 2>&1 | grep -i "first report\|FAILED\|passed\|failed" | cut -c1-300 | head -5
done
