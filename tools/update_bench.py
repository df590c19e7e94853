"""update() of BASELINE configs 2 / 3 / 4 / 5 (row f4): one training step -- loss forward + backward, gradient-norm clip, AdamW, EMA -- on
the library's nodes (default) against the reference's ATen autograd graph (CDX_TRAIN_NATIVE=0), same box, same batch.
Usage (GPU box): python tools/update_bench.py [cfg1 cfg2 cfg3 cfg4 cfg5 chitf sfbc]   -> one line per (config, mode): ms per update()."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.diffusion import ContinuousDiffusionSDE, ContinuousEDM, DiscreteDiffusionSDE  # noqa: E402
from cleandiffuser_amd.diffusion.ddpm import DDPM  # noqa: E402
from cleandiffuser_amd.nn_condition import IdentityCondition, MLPCondition  # noqa: E402
from cleandiffuser_amd.nn_diffusion import ChiTransformer, ChiUNet1d, DiT1d, IDQLMlp, JannerUNet1d, PearceMlp, SfBCUNet  # noqa: E402
from cleandiffuser_amd.utils import load_synth  # noqa: E402

DEV = torch.device("cuda", 0)


def build(name):
    if name == "cfg2":
        net = load_synth(JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
        fix = torch.zeros(32, 23)
        fix[0, :17] = 1.0
        agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, diffusion_steps=20, predict_noise=False, grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 32, 23, device=DEV), None, "config 2: JannerUNet1d H=32 D=23, batch 256"
    elif name == "cfg3":
        net = load_synth(ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2], obs_as_global_cond=True))
        agent = DDPM(net, IdentityCondition(dropout=0.0), diffusion_steps=50, grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 16, 2, device=DEV).clamp(-1, 1), torch.randn(256, 2, 20, device=DEV), "config 3: ChiUNet1d 68.9 M parameters, batch 256"
    elif name == "cfg4":
        net = load_synth(DiT1d(29, emb_dim=128, d_model=320, n_heads=10, depth=2, timestep_emb_type="fourier"))
        cnd = load_synth(MLPCondition(1, 128, [128], torch.nn.SiLU(), dropout=0.25), 2)
        agent = ContinuousDiffusionSDE(net, cnd, predict_noise=True, noise_schedule="linear", grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(64, 64, 29, device=DEV), torch.rand(64, 1, device=DEV), "config 4: DiT1d d=320 h=10 depth=2, 64 tokens, batch 64"
    elif name == "cfg1":
        net = load_synth(PearceMlp(6, To=1, emb_dim=128, hidden_dim=512))
        agent = DDPM(net, IdentityCondition(dropout=0.0), diffusion_steps=100, grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 6, device=DEV).clamp(-1, 1), torch.randn(256, 1, 128, device=DEV), "config 1: PearceMlp act=6, hidden 512, batch 256"
    elif name == "sfbc":
        net = load_synth(SfBCUNet(6, emb_dim=64))
        agent = ContinuousDiffusionSDE(net, IdentityCondition(dropout=0.0), predict_noise=True, noise_schedule="linear", grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 6, device=DEV), torch.randn(256, 64, device=DEV), "sfbc: SfBCUNet act=6 (512, 256, 128), batch 256"
    elif name == "chitf":
        net = load_synth(ChiTransformer(2, 5, 10, 2, d_model=256, nhead=4, num_layers=8, p_drop_attn=0.3))
        agent = DDPM(net, IdentityCondition(dropout=0.0), diffusion_steps=50, grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 10, 2, device=DEV).clamp(-1, 1), torch.randn(256, 2, 5, device=DEV), \
            "dp_pusht transformer: ChiTransformer d=256 h=4 x 8 layers, Ta 10 / To 2, attention dropout 0.3, batch 256"
    else:
        net = load_synth(IDQLMlp(0, 15, emb_dim=128, hidden_dim=1024, n_blocks=6, dropout=0.1))
        agent = ContinuousEDM(net, None, grad_clip_norm=1.0, device=DEV)
        x0, cond, what = torch.randn(256, 15, device=DEV), None, "config 5: IDQLMlp 1024 x 6 (dropout 0.1), batch 256"
    agent.train()
    return agent, x0, cond, what


def main():
    for name in (sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "chitf", "sfbc"]):
        res = {}
        modes = ("graph",) if os.environ.get("UPDATE_BENCH_GRAPH_ONLY") else ("graph", "graph_nosplitk", "native", "aten")
        for mode in modes:
            os.environ["CDX_TRAIN_NATIVE"] = "0" if mode == "aten" else "1"
            os.environ["CDX_TRAIN_GRAPH"] = "auto" if mode.startswith("graph") else "0"
            os.environ["CDX_TRAIN_SPLITK"] = "0" if mode == "graph_nosplitk" else "1"
            agent, x0, cond, what = build(name)
            call = (lambda: agent.update(x0, cond)) if cond is not None else (lambda: agent.update(x0))
            for _ in range(4):
                call()
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                log = call()
            torch.cuda.synchronize()
            res[mode] = 1e3 * (time.perf_counter() - t0) / reps
            assert float(log["loss"]) == float(log["loss"])
            if mode.startswith("graph"):
                assert agent.__dict__.get("_cdx_graphed"), agent.__dict__.get("_cdx_graph_off")
        if len(modes) == 1:
            print(f"{what}: update() {res['graph']:.3f} ms by default", flush=True)
            continue
        print(f"{what}: update() {res['graph']:.2f} ms by default (the library's nodes, forward + backward replayed as one HIP graph), "
              f"{res['graph_nosplitk']:.2f} ms without split-K scratch for the small-batch GEMMs, "
              f"{res['native']:.2f} ms eager on the library's nodes, {res['aten']:.2f} ms on ATen autograd "
              f"({res['aten'] / res['graph']:.2f}x)", flush=True)


if __name__ == "__main__":
    main()
