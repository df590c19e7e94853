"""GPU box: GroupNorm(8 groups) -> Mish backward on (B, 32, 32) inputs three ways -- the library's cdx_groupnorm_bwd_f32, ATen on the device,
ATen on the CPU in float64 / float32.  The record profiles/r04_aten_groupnorm_backward.txt came from this script: ATen's device kernel of this
ROCm build returns gain / shift gradients that are off by 100 % once the batch reaches 255."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.engine import blocks
import torch.nn.functional as F
DEV = "cuda:0"
for B in (128, 255, 256, 257, 512):
    C, L = 32, 32
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * L, C, generator=g); dy = torch.randn(B * L, C, generator=g)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)); beta = (0.1 * torch.randn(C, generator=g))
    dx, dg, db = blocks.groupnorm_backward(dy.to(DEV), x.to(DEV), gamma.to(DEV), beta.to(DEV), B, L, 8, act="mish", param_grads=True)
    outs = {}
    for dev, dt in (("cuda:0", torch.float32), ("cpu", torch.float64), ("cpu", torch.float32)):
        xr = x.to(dev, dt).view(B, L, C).permute(0, 2, 1).clone().requires_grad_(True); gr = gamma.to(dev, dt).clone().requires_grad_(True); br = beta.to(dev, dt).clone().requires_grad_(True)
        y = F.mish(F.group_norm(xr, 8, gr, br, 1e-5))
        y.backward(dy.to(dev, dt).view(B, L, C).permute(0, 2, 1))
        outs[(dev, dt)] = (gr.grad.double().cpu(), br.grad.double().cpu())
    ref = outs[("cpu", torch.float64)]
    print("B", B, "| native vs fp64:", float((dg.double().cpu() - ref[0]).abs().max()), float((db.double().cpu() - ref[1]).abs().max()),
          "| ATen-GPU vs fp64:", float((outs[("cuda:0", torch.float32)][0] - ref[0]).abs().max()), float((outs[("cuda:0", torch.float32)][1] - ref[1]).abs().max()),
          "| ATen-CPU fp32 vs fp64:", float((outs[("cpu", torch.float32)][0] - ref[0]).abs().max()), "| scale", float(ref[0].abs().max()))
