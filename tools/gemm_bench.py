"""Micro-benchmark of cdx_gemm_f32 on the GEMM shapes of configs 4 and 5 (GPU box):  python tools/gemm_bench.py
Prints one line per shape: time per launch (HIP events over `reps` launches), TFLOP/s, fraction of the 157.3 TF fp32-MFMA peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.engine import blocks  # noqa: E402

SHAPES = [  # (M, N, K, act, gate/residual)
    ("dit qkv", 65536, 960, 320, "none", False),
    ("dit out_proj", 65536, 320, 320, "none", True),
    ("dit fc1", 65536, 1280, 320, "gelu_tanh", False),
    ("dit fc2", 65536, 320, 1280, "none", True),
    ("dit adaLN", 306, 1920, 320, "none", False),
    ("dit final", 19584, 29, 320, "none", False),
    ("dit x_proj", 9792, 320, 29, "none", False),
    ("mlp fc1", 16384, 4096, 1024, "mish", False),
    ("mlp fc2", 16384, 1024, 4096, "none", True),
    ("mlp fc1 big", 131072, 4096, 1024, "mish", False),
]


def main(reps=20):
    dev = "cuda:0"
    shapes = SHAPES
    if len(sys.argv) > 1:                                  # custom shapes: M,N,K[,act[,gr]] ...
        shapes = []
        for spec in sys.argv[1:]:
            f = spec.split(",")
            shapes.append((spec, int(f[0]), int(f[1]), int(f[2]), f[3] if len(f) > 3 else "none", len(f) > 4))
    for name, m, n, k, act, gr in shapes:
        a = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        out = torch.empty(m, n, device=dev)
        kw = dict(act=act)
        if gr:
            kw.update(gate=torch.randn((m + 63) // 64, n, device=dev), rows_per_gate=64, residual=torch.randn(m, n, device=dev))
        for _ in range(3):
            blocks.linear(a, w, b, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            blocks.linear(a, w, b, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tf = 2.0 * m * n * k / us / 1e6
        print(f"{name:14s} M={m:6d} N={n:5d} K={k:5d} {us:9.1f} us  {tf:6.1f} TF  {tf / 157.3:5.1%}", flush=True)


if __name__ == "__main__":
    main()
