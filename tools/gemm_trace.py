"""Workgroup timeline of one cdx_gemm_f32 launch (s_memtime stamps, 100 MHz): where a tile's life goes.
Usage (GPU box): python tools/gemm_trace.py M N K [act]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.engine import blocks  # noqa: E402
from cleandiffuser_amd.engine.runtime import load_library  # noqa: E402


def main(m, n, k, act="none"):
    dev = "cuda:0"
    lib = load_library()
    lib.cdx_gemm_set_trace.argtypes = [ctypes.c_void_p]
    a, w, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    buf = torch.zeros(tiles * 4, dtype=torch.int64, device=dev)
    blocks.linear(a, w, b, out=out, act=act)
    torch.cuda.synchronize()
    lib.cdx_gemm_set_trace(buf.data_ptr())
    blocks.linear(a, w, b, out=out, act=act)
    torch.cuda.synchronize()
    lib.cdx_gemm_set_trace(None)
    t = buf.cpu().numpy().reshape(tiles, 4).astype(np.float64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                      # 100 MHz -> microseconds
    print(f"M={m} N={n} K={k} act={act}: {tiles} workgroups, kernel span {us.max():.1f} us")
    ph = np.diff(us, axis=1)
    for name, col in (("stage first tile", 0), ("K loop", 1), ("epilogue", 2)):
        print(f"  {name:18s} mean {ph[:, col].mean():7.2f} us   p10 {np.percentile(ph[:, col], 10):7.2f}   p90 {np.percentile(ph[:, col], 90):7.2f}")
    order = np.argsort(us[:, 0])
    starts = us[order, 0]
    print("  start times (us) deciles:", np.round(np.percentile(starts, range(0, 101, 10)), 1))
    print("  end times   (us) deciles:", np.round(np.percentile(us[:, 3], range(0, 101, 10)), 1))
    # phase concurrency inside XCD 0 (blockIdx % 8 == 0 share one clock): how many workgroups are in their K loop /
    # epilogue at the same instant?  Lockstep shows up as the epilogue count swinging between 0 and "all slots".
    x = t[0::8]
    x = x - x[:, 0].min()
    grid_t = np.linspace(0, x[:, 3].max(), 60)
    in_k = [(int(((x[:, 1] <= g) & (g < x[:, 2])).sum()), int(((x[:, 2] <= g) & (g < x[:, 3])).sum())) for g in grid_t]
    print("  XCD0 (K-loop, epilogue) workgroups over time:", " ".join(f"{a}/{b}" for a, b in in_k))
    for i in (0, tiles // 2, tiles - 1):
        print(f"  wg {i:5d}: start {us[i, 0]:7.1f} staged {us[i, 1]:7.1f} kdone {us[i, 2]:7.1f} end {us[i, 3]:7.1f}")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), *(sys.argv[4:5]))
