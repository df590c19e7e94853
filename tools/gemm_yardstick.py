"""Yardstick only (never on the product path): the vendor fp32 GEMM (torch.nn.functional.linear -> hipBLASLt/rocBLAS)
on the same shapes as tools/gemm_bench.py, to know what the silicon sustains in fp32 MFMA."""
import torch
import torch.nn.functional as F

from gemm_bench import SHAPES

for name, m, n, k, act, gr in SHAPES:
    a = torch.randn(m, k, device="cuda:0")
    w = torch.randn(n, k, device="cuda:0")
    b = torch.randn(n, device="cuda:0")
    for _ in range(3):
        F.linear(a, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        F.linear(a, w, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tf = 2.0 * m * n * k / us / 1e6
    print(f"{name:14s} M={m:6d} N={n:5d} K={k:5d} {us:9.1f} us  {tf:6.1f} TF  {tf / 157.3:5.1%}  (vendor, bias only)", flush=True)
