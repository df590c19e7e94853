#!/bin/bash
mkdir -p gpurun_out/r4h
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "native_training or update_runs" > gpurun_out/r4h/pytest.log 2>&1
tail -12 gpurun_out/r4h/pytest.log
python tools/aten_groupnorm_backward_check.py > gpurun_out/r4h/aten_groupnorm_backward.txt 2>&1
timeout 300 python - > gpurun_out/r4h/update_profile.txt 2>&1 <<'PY'
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_configs as bc
from torch.profiler import profile, ProfilerActivity
for native in (True, False):
    label, call, b = bc.cfgU(256, native_backward=native)
    for _ in range(3): call()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5): call()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    tot = sum(e.self_device_time_total for e in ka) / 5 / 1e3
    print(f"native={native}: device time per update {tot:.3f} ms; top kernels:")
    for e in sorted(ka, key=lambda e: -e.self_device_time_total)[:12]:
        print(f"   {e.self_device_time_total/5/1e3:8.3f} ms  x{e.count//5:4d}  {e.key[:110]}")
    if native:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20): call()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
PY
head -90 gpurun_out/r4h/update_profile.txt
