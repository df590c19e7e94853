#!/bin/bash
mkdir -p gpurun_out/r4n
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "graphed_update or native_training or update_runs" > gpurun_out/r4n/pytest.log 2>&1
tail -15 gpurun_out/r4n/pytest.log
timeout 300 python - > gpurun_out/r4n/update_bench.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_configs as bc
for B in (64, 256):
  for native, graph in ((True, True), (True, False), (False, False)):
    label, call, b = bc.cfgU(B, native_backward=native, graph=graph)
    for _ in range(4): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"B={B} native={native} graph={graph}: {1e3*dt:.3f} ms per update() ({1/dt:.1f} steps/s)", flush=True)
PY
cat gpurun_out/r4n/update_bench.txt
