#!/bin/bash
# Flake hunt: the first tests of the GPU suite (split / grouped self-checks happen there) in N fresh processes, full tracebacks kept.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4stress
for i in $(seq 1 ${N:-10}); do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=long -k "test_fused_sample_matches_reference_fixture or small_batch or split or group" > gpurun_out/r4stress/run$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/r4stress/run$i.log)"
done
grep -l "FAILED\|failed" gpurun_out/r4stress/*.log | head
