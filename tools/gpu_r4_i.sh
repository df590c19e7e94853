#!/bin/bash
mkdir -p gpurun_out/r4i
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "native_training or update_runs or loss_and_update" > gpurun_out/r4i/pytest.log 2>&1
tail -5 gpurun_out/r4i/pytest.log
timeout 300 python - > gpurun_out/r4i/update_bench.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_configs as bc
for B in (64, 256):
  for native in (True, False):
    label, call, b = bc.cfgU(B, native_backward=native)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"B={B} native={native}: {1e3*dt:.3f} ms per update() ({1/dt:.1f} steps/s)")
PY
cat gpurun_out/r4i/update_bench.txt
