import sys, torch
sys.path.insert(0, "/root/repo")
from cleandiffuser_amd.engine import blocks
for (m, n, k) in [(4096, 3072, 16384), (4096, 2048, 16384), (4096, 1024, 16384), (8192, 3072, 8192)]:
    a = torch.randn(m, k, device="cuda:0"); w = torch.randn(n, k, device="cuda:0") / k ** 0.5
    out = torch.empty(m, n, device="cuda:0")
    for _ in range(2): blocks.linear(a, w, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): blocks.linear(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    tiles = (m // 128) * (n // 128)
    print(f"M={m} N={n} K={k} tiles={tiles} {us:9.1f} us {2.0*m*n*k/us/1e6:6.1f} TF", flush=True)
    import torch.nn.functional as F
    for _ in range(2): F.linear(a, w)
    e0.record()
    for _ in range(5): F.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f"   vendor {us:9.1f} us {2.0*m*n*k/us/1e6:6.1f} TF", flush=True)
