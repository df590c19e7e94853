"""Per-op cycle breakdown of the fused kernel (workgroup 0, first forward) -- debug/tuning aid.
Usage (GPU box): python tools/op_profile.py [batch] > gpurun_out/op_profile.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleandiffuser_amd.engine import program as P, runtime  # noqa: E402


os.environ.setdefault("CDX_UNET2", "0")      # this tool stamps the FIRST program kernel (cdx_unet1d_kernel); tools/op_profile2.py is the v2 one


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    model_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    brief = len(sys.argv) > 3
    bench.BATCH = batch
    dev = torch.device("cuda", 0)
    if model_dim == 32:
        agent, net = bench.build_agent(dev)
    else:   # same topology, narrower channels: the whole weight set becomes L2-resident
        from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE
        from cleandiffuser_amd.nn_diffusion import JannerUNet1d
        from cleandiffuser_amd.utils import load_synth
        net = load_synth(JannerUNet1d(23, model_dim=model_dim, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 0)
        fm = torch.zeros(32, 23)
        fm[0, :17] = 1
        agent = DiscreteDiffusionSDE(net, None, fix_mask=fm, diffusion_steps=20, predict_noise=False, device=dev)
        agent.eval()
    prior, z0 = bench.make_inputs(dev, 0)
    kw = dict(solver="ddim", n_samples=batch, sample_steps=20, temperature=0.5)
    for _ in range(3):
        agent.sample(prior, noise=[z0], **kw)
    prog = runtime.compiled_program(agent.model_ema["diffusion"], 32).prog
    n_ops = len(prog.ops)
    buf = torch.zeros(n_ops * 8 + 2, dtype=torch.int64, device=dev)
    runtime.set_profile_buffer(buf)
    agent.sample(prior, noise=[z0], **kw)
    torch.cuda.synchronize()
    runtime.set_profile_buffer(None)
    t = buf.cpu().numpy()
    total = t[n_ops * 8 + 1] - t[n_ops * 8]
    print(f"batch={batch} kernel cycles (wg0) = {total}  lds_bytes={prog.lds_floats * 4}")
    fwd = t[(n_ops - 1) * 8 + 3] - t[0]
    print(f"first forward cycles = {fwd}  ({fwd * 20 / total:.2%} of kernel if all 20 equal)")
    tot_k = tot_s = tot_e = 0
    if brief:
        ks = [t[i * 8 + 1] - t[i * 8] for i, op in enumerate(prog.ops) if op[P.W_KIND] == P.OP_CONV]
        es = [t[i * 8 + 3] - t[i * 8 + 2] for i, op in enumerate(prog.ops) if op[P.W_KIND] == P.OP_CONV]
        print("kloop cycles per conv:", ks)
        print("epilogue cycles per conv:", es)
        return
    print(f"{'op':>3} {'kind':>6} {'cout':>4} {'L':>3} {'taps':>4} {'chunks':>6} {'ks':>2} {'items':>5} {'kloop':>7} {'sync':>6} {'epi':>6} {'total':>7}")
    for i, op in enumerate(prog.ops):
        s0, s1, s2, s3, k1, k2, k3, k4 = t[i * 8:i * 8 + 8]
        if op[P.W_KIND] == P.OP_CONV:
            k, s, e = s1 - s0, s2 - s1, s3 - s2
            tot_k, tot_s, tot_e = tot_k + k, tot_s + s, tot_e + e
            items = (op[P.W_COUT16] // 16) * op[P.W_KSPLIT]
            print(f"{i:3d} {'conv':>6} {op[P.W_COUT]:4d} {op[P.W_LOUT]:3d} {op[P.W_TAPS]:4d} {op[P.W_NCHUNKS]:6d} "
                  f"{op[P.W_KSPLIT]:2d} {items:5d} {k:7d} {s:6d} {e:6d} {s3 - s0:7d} | item {k1 - s0:5d} "
                  f"operands {k2 - k1:5d} mfma {k3 - k2:6d} tail {k4 - k3:5d} pre-barrier {s1 - k4:5d}")
        else:
            print(f"{i:3d} {'lin' if op[P.W_KIND] == P.OP_LINEAR else 'temb':>6} {'':4} {'':3} {'':4} {'':6} {'':2} {'':5} {'':7} {'':6} {'':6} {s3 - s0:7d}")
    print(f"conv totals: kloop={tot_k} sync={tot_s} epilogue={tot_e}  sum={tot_k + tot_s + tot_e}")


if __name__ == "__main__":
    main()
