"""DiT1d error budget (VERDICT r3 'weak' #2, 'next' #1 iii): where the distance between the native DiT path and the reference fixture
comes from.  Run on a GPU box: ``python tools/dit_error_budget.py > gpurun_out/dit_error_budget.txt``.

Part A -- op-local error.  One forward of the config-4 network (d 320, 10 heads, depth 2, 64 tokens, B = 3) is evaluated op by op in
float64 on the CPU (the yardstick); every op is then re-evaluated from the SAME (float64 -> float32 rounded) inputs (a) by the native
kernel that serves it on the device (cdx_gemm_f32 with its fused epilogue, cdx_layernorm_f32, cdx_attention_f32) and (b) by the ATen
fp32 CPU op the reference would run.  Reported: max |err| against the float64 output, in units of the output's rms -- the rounding
each implementation adds by itself, before anything is amplified.

Part B -- end to end.  The fixture scenarios are sampled three ways from the same weights and draws: natively on the device, by this
package's PyTorch executor in fp32 on the CPU (equal to the imported reference at 2e-6, tests/test_extra_fixtures.py) and by the same
executor in float64.  |fp32 CPU - fp64| is the reference's OWN rounding noise on the scenario; |native - fp64| is ours; the parity
bar compares the two fp32 results with each other, i.e. the sum of two such noises.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from cleandiffuser_amd.engine import blocks                       # noqa: E402
from cleandiffuser_amd.utils import load_synth                    # noqa: E402
from oracle import cases, extra_cases                             # noqa: E402  (test infrastructure: this is a measurement tool)

DEV = "cuda:0"


def rel(a, ref):
    ref = ref.double().cpu()
    return float((a.double().cpu() - ref).abs().max() / ref.pow(2).mean().sqrt())


def part_a():
    lib = cases.lib_namespace("amd")
    net = load_synth(lib.DiT1d(29, emb_dim=128, d_model=320, n_heads=10, depth=2, timestep_emb_type="fourier"), 56)
    net64 = load_synth(lib.DiT1d(29, emb_dim=128, d_model=320, n_heads=10, depth=2, timestep_emb_type="fourier"), 56).double()
    g = torch.Generator().manual_seed(5)
    B, T, d, H = 3, 64, 320, 10
    x = (0.5 * torch.randn(B, T, 29, generator=g)).double()
    t = torch.tensor([0.7, 0.4, 0.9], dtype=torch.float64)
    cond = (0.3 * torch.randn(B, 128, generator=g)).double()
    rows = []

    def both(name, out64, native, aten):
        rows.append((name, rel(native, out64), rel(aten, out64), float(out64.pow(2).mean().sqrt())))

    def dev(v):
        return v.float().to(DEV).contiguous()

    with torch.no_grad():
        emb = net64.map_emb(net64.map_noise(t) + cond)                                  # (B, d) float64
        h = net64._tokens(x)                                                            # (B, T, d)
        w = net.x_proj
        pos = net64.pos_emb_cache.float()
        both("x_proj + pos table (K = 29)", h.reshape(B * T, d),
             blocks.linear(dev(x.reshape(B * T, 29)), dev(w.weight), dev(w.bias), table=dev(pos)),
             F.linear(x.float().reshape(B * T, 29), w.weight, w.bias) + pos.repeat(B, 1))
        for bi, (blk, blk64) in enumerate(zip(net.blocks, net64.blocks)):
            mod = blk64.adaLN_modulation(emb)                                           # (B, 6 d)
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
            hin = h.reshape(B * T, d)
            ln1 = F.layer_norm(h, (d,), eps=1e-6) * (1 + sc_a[:, None]) + sh_a[:, None]
            both(f"block {bi}: LayerNorm + modulate", ln1.reshape(B * T, d),
                 blocks.layernorm(dev(hin), scale=dev(sc_a), shift=dev(sh_a), rows_per_mod=T, eps=1e-6),
                 (F.layer_norm(h.float(), (d,), eps=1e-6) * (1 + sc_a.float()[:, None]) + sh_a.float()[:, None]).reshape(B * T, d))
            a64, a32 = blk64.attn, blk.attn
            qkv = F.linear(ln1, a64.in_proj_weight, a64.in_proj_bias)                   # (B, T, 3 d)
            both(f"block {bi}: qkv GEMM (K = 320)", qkv.reshape(B * T, 3 * d),
                 blocks.linear(dev(ln1.reshape(B * T, d)), dev(a32.in_proj_weight), dev(a32.in_proj_bias)),
                 F.linear(ln1.float().reshape(B * T, d), a32.in_proj_weight, a32.in_proj_bias))
            q, k, v = (z.reshape(B, T, H, d // H).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
            att = torch.softmax(q @ k.transpose(-1, -2) / (d // H) ** 0.5, dim=-1) @ v  # (B, H, T, dh)
            att = att.transpose(1, 2).reshape(B * T, d)
            q32, k32, v32 = (z.float() for z in (q, k, v))
            both(f"block {bi}: attention core (softmax(QK^T / sqrt dh) V)", att,
                 blocks.attention(dev(qkv.reshape(B * T, 3 * d)), B, T, H),
                 (torch.softmax(q32 @ k32.transpose(-1, -2) / (d // H) ** 0.5, dim=-1) @ v32).transpose(1, 2).reshape(B * T, d))
            o = F.linear(att, a64.out_proj.weight, a64.out_proj.bias).reshape(B, T, d)
            h1 = ln1 + g_a[:, None] * o
            both(f"block {bi}: out_proj GEMM * gate + residual", h1.reshape(B * T, d),
                 blocks.linear(dev(att), dev(a32.out_proj.weight), dev(a32.out_proj.bias), gate=dev(g_a), rows_per_gate=T,
                               residual=dev(ln1.reshape(B * T, d))),
                 (ln1.float() + g_a.float()[:, None] * F.linear(att.float(), a32.out_proj.weight, a32.out_proj.bias).reshape(B, T, d)).reshape(B * T, d))
            ln2 = F.layer_norm(h1, (d,), eps=1e-6) * (1 + sc_m[:, None]) + sh_m[:, None]
            f1 = F.gelu(F.linear(ln2, blk64.mlp[0].weight, blk64.mlp[0].bias), approximate="tanh")
            both(f"block {bi}: fc1 GEMM + GELU(tanh)", f1.reshape(B * T, 4 * d),
                 blocks.linear(dev(ln2.reshape(B * T, d)), dev(blk.mlp[0].weight), dev(blk.mlp[0].bias), act="gelu_tanh"),
                 F.gelu(F.linear(ln2.float().reshape(B * T, d), blk.mlp[0].weight, blk.mlp[0].bias), approximate="tanh"))
            f2 = F.linear(f1, blk64.mlp[3].weight, blk64.mlp[3].bias)
            h = h1 + g_m[:, None] * f2
            both(f"block {bi}: fc2 GEMM (K = 1280) * gate + residual", h.reshape(B * T, d),
                 blocks.linear(dev(f1.reshape(B * T, 4 * d)), dev(blk.mlp[3].weight), dev(blk.mlp[3].bias), gate=dev(g_m), rows_per_gate=T,
                               residual=dev(h1.reshape(B * T, d))),
                 (h1.float() + g_m.float()[:, None] * F.linear(f1.float(), blk.mlp[3].weight, blk.mlp[3].bias)).reshape(B * T, d))
        fl, fl64 = net.final_layer, net64.final_layer
        shift, scale = fl64.adaLN_modulation(emb).chunk(2, dim=1)
        lnf = F.layer_norm(h, (d,), eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
        out = F.linear(lnf, fl64.linear.weight, fl64.linear.bias)
        both("final Linear (N = 29)", out.reshape(B * T, 29),
             blocks.linear(dev(lnf.reshape(B * T, d)), dev(fl.linear.weight), dev(fl.linear.bias)),
             F.linear(lnf.float().reshape(B * T, d), fl.linear.weight, fl.linear.bias))
        full64 = net64._forward_torch(x, t, cond)
        full_native = net.to(DEV)(x.float().to(DEV), t.float().to(DEV), cond.float().to(DEV))
        net.cpu()
        full_aten = net._forward_torch(x.float(), t.float(), cond.float())
        both("WHOLE forward (one network evaluation)", full64.reshape(B * T, 29), full_native.reshape(B * T, 29), full_aten.reshape(B * T, 29))
    print("Part A: one forward of the config-4 DiT1d, op-local error = max |y - y64| / rms(y64), inputs identical (float64 rounded to float32)")
    print(f"{'op':58s} {'native (MI355X)':>16s} {'ATen fp32 (CPU)':>16s} {'rms(y64)':>10s}")
    for name, en, ea, rms in rows:
        print(f"{name:58s} {en:16.3e} {ea:16.3e} {rms:10.3f}")


def _to_fp64(agent):
    agent.model.double()
    agent.model_ema.double()
    for holder in (agent.model, agent.model_ema):          # (the solver builds its timestep vector in float32)
        net = holder["diffusion"]
        fwd = net.forward
        net.forward = (lambda f: lambda x, noise, condition=None: f(x, noise.double(), condition))(fwd)
    for k, v in list(vars(agent).items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(agent, k, v.double())


def part_b():
    print("\nPart B: whole sample() calls, max |difference| over the result (same weights, same draws)")
    print(f"{'scenario':28s} {'|x|max':>8s} {'native - fp64':>14s} {'CPU fp32 - fp64':>16s} {'native - CPU fp32':>18s} {'native - fixture':>17s}")
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    # (1) the config-4 network at its exact setting, output layer tied (|x| <= 6.2); (2) the same with plain synthetic weights
    for name in ("baseline_cfg4_tied", "baseline_cfg4"):
        nat = extra_cases.run(name, "amd", DEV)["x"].double().cpu()
        c32 = extra_cases.run(name, "amd", "cpu")["x"].double()
        c64 = extra_cases.run(name, "amd", "cpu", fp64=True)["x"].double()
        fix = torch.from_numpy(np.load(os.path.join(golden, f"extra_{name}.npz"))["x"]).double()
        print(f"{name:28s} {float(fix.abs().max()):8.2f} {float((nat - c64).abs().max()):14.3e} {float((c32 - c64).abs().max()):16.3e} "
              f"{float((nat - c32).abs().max()):18.3e} {float((nat - fix).abs().max()):17.3e}")
    # (3) the smoke fixture (small DiT1d, clipped)
    name = "dit_cfg4_cfg2_dpmpp2m"
    lib = cases.lib_namespace("amd")
    inp = cases.make_inputs(name)
    outs = {}
    for tag, dev, dbl in (("nat", DEV, False), ("c32", "cpu", False), ("c64", "cpu", True)):
        agent, _ = cases.build(lib, name, device=dev)
        kw = cases.sample_kwargs(name, inp, device=dev)
        kw["noise"] = [torch.from_numpy(z).to(dev) for z in inp["noise"][:1]]      # (an ODE solver: the initial draw only)
        prior = torch.from_numpy(inp["prior"]).to(dev)
        if dbl:
            _to_fp64(agent)
            prior = prior.double()
            kw = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else
                      [z.double() for z in v] if isinstance(v, list) else v) for k, v in kw.items()}
            torch.set_default_dtype(torch.float64)
        try:
            x, _ = agent.sample(prior, **kw)
        finally:
            torch.set_default_dtype(torch.float32)
        outs[tag] = x.double().cpu()
    fix = torch.from_numpy(np.load(os.path.join(golden, f"{name}.npz"))["x_out"]).double()
    print(f"{name:28s} {float(fix.abs().max()):8.2f} {float((outs['nat'] - outs['c64']).abs().max()):14.3e} "
          f"{float((outs['c32'] - outs['c64']).abs().max()):16.3e} {float((outs['nat'] - outs['c32']).abs().max()):18.3e} "
          f"{float((outs['nat'] - fix).abs().max()):17.3e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    part_a()
    part_b()
