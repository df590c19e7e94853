import sys, json, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_configs as bc
from cleandiffuser_amd.engine import classifier_grad
for B in (256, 3200):
    label, call, _, _ = bc.cfg2g(B)
    for mode in ("native", "autograd"):
        if mode == "autograd":
            orig = classifier_grad.gradients
            classifier_grad.gradients = lambda *a, **k: None
        call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"B={B} {mode}: {1e3*dt:.1f} ms per sample() -> {B/dt:.0f} trajectories/s", flush=True)
        if mode == "autograd":
            classifier_grad.gradients = orig
