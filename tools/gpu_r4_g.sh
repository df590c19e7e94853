#!/bin/bash
mkdir -p gpurun_out/r4g
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "native_training or update_runs or loss_and_update or fused_adamw or full_size_properties or float64_yardstick" > gpurun_out/r4g/pytest.log 2>&1
tail -40 gpurun_out/r4g/pytest.log
timeout 300 python - > gpurun_out/r4g/update_bench.txt 2>&1 <<'PY'
import os, sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_configs as bc
for native in (True, False):
    label, call, b = bc.cfgU(256, native_backward=native)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"native={native}: {1e3*dt:.3f} ms per update() at B=256 ({1/dt:.1f} steps/s)")
PY
cat gpurun_out/r4g/update_bench.txt
timeout 200 python tools/time_cfg2.py 256 32 128 > gpurun_out/r4g/time.txt 2>&1; cat gpurun_out/r4g/time.txt
