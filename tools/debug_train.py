import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.nn_diffusion import JannerUNet1d
from cleandiffuser_amd.utils import load_synth
DEV = "cuda:0"
net = load_synth(JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 7).to(DEV)
for B in (24, 64, 128, 256):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 32, 23, generator=g).to(DEV)
    t = torch.randint(0, 20, (B,), generator=g).to(DEV)
    wgt = torch.randn(B, 32, 23, generator=g).to(DEV)
    res = {}
    for native in (True, False):
        os.environ["CDX_TRAIN_NATIVE"] = "1" if native else "0"
        net.zero_grad(set_to_none=True)
        y = net(x, t, None)
        ((y * wgt).sum() / B).backward()
        res[native] = {n: p.grad.clone() for n, p in net.named_parameters()}
    worst = sorted(((float((res[True][n] - res[False][n]).abs().max()) / (float(res[False][n].abs().max()) + 1e-12), n) for n in res[True]), reverse=True)[:6]
    gn = {k: float(torch.sqrt(sum((v ** 2).sum() for v in res[k].values()))) for k in res}
    print("B", B, "grad norms", gn, "worst", [(f"{e:.2e}", n) for e, n in worst])
