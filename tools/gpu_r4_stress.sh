#!/bin/bash
# Flake hunt: the split / grouped tests + B = 32 / 256 timing loops in N fresh processes on whatever box this call lands on.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4stress
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
for i in $(seq 1 ${N:-4}); do
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -rs --tb=short -W always -k "test_fused_sample_matches_reference_fixture or small_batch or split or group" > gpurun_out/r4stress/run$i.log 2>&1
  echo "run $i: $(tail -1 gpurun_out/r4stress/run$i.log)"; grep -n "first report\|FAILED" gpurun_out/r4stress/run$i.log | cut -c1-300 | head -3
done
timeout 300 python tools/host_profile.py 32 2>&1 | sed -n 2,2p
timeout 300 python tools/time_cfg2.py 256 32 64 2>&1 | grep -v amdgpu | cut -c1-200
