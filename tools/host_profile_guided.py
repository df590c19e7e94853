import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import bench_configs as bc
for mode in ("1", "0"):
    os.environ["CDX_UNET2_GUIDED_GROUP"] = mode
    label, call, b, flops = bc.cfg2g(B=256)
    for _ in range(3): call()
    torch.cuda.synchronize()
    hs = []
    t0 = time.perf_counter()
    for _ in range(10):
        h0 = time.perf_counter(); call(); hs.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    print(f"GUIDED_GROUP={mode}: wall {1e3*wall:.3f} ms per call; host time of a call (enqueue only) min {1e3*min(hs):.3f} median {1e3*sorted(hs)[5]:.3f} max {1e3*max(hs):.3f} ms")
    if mode == "1":
        pr = cProfile.Profile(); pr.enable()
        for _ in range(5): call()
        pr.disable(); torch.cuda.synchronize()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
