"""Per-op cycle breakdown of a GUIDED v2 program (denoiser ops, then the classifier's forward + backward ops), workgroup 0, second
step.  Default: the config-2 guided program (H = 32, dim_mult (1, 2, 2, 2); 131 KB of LDS + the stamp buffer); `h16`: the same
architecture one level shallower.  Usage: python tools/op_profile2_guided.py [batch] [h16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.classifier import CumRewClassifier  # noqa: E402
from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE  # noqa: E402
from cleandiffuser_amd.engine import program2 as P2, runtime, runtime2  # noqa: E402
from cleandiffuser_amd.nn_classifier import HalfJannerUNet1d  # noqa: E402
from cleandiffuser_amd.nn_diffusion import JannerUNet1d  # noqa: E402
from cleandiffuser_amd.utils import load_synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    H, D, dm = (16, 23, [1, 2, 2]) if "h16" in sys.argv[2:] else (32, 23, [1, 2, 2, 2])
    net = load_synth(JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=dm, kernel_size=5))
    clf_net = load_synth(HalfJannerUNet1d(H, D, out_dim=1, model_dim=32, emb_dim=32, dim_mult=tuple(dm), kernel_size=3), 1)
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, classifier=CumRewClassifier(clf_net, device=dev), diffusion_steps=20,
                                 predict_noise=False, device=dev)
    agent.eval()
    prior = torch.zeros(B, H, D, device=dev)
    prior[:, 0, :17] = torch.randn(B, 17, device=dev)
    call = lambda: agent.sample(prior, solver="ddpm", n_samples=B, sample_steps=20, temperature=0.5, w_cg=0.1)[0]  # noqa: E731
    orig_logp = agent.classifier.logp

    def logp(*a, **k):                      # the final log p call runs on the first program kernel: keep it off the stamp buffer
        saved, runtime._prof["buf"] = runtime._prof["buf"], None
        try:
            return orig_logp(*a, **k)
        finally:
            runtime._prof["buf"] = saved
    agent.classifier.logp = logp
    for _ in range(3):
        call()
    comp = runtime2.compiled_guided2(agent.model_ema["diffusion"], agent.classifier.model_ema, H)
    prog = comp.prog
    n_ops = len(prog.ops)
    buf = torch.zeros(n_ops * 8 + 2, dtype=torch.int64, device=dev)
    runtime.set_profile_buffer(buf)
    call()
    torch.cuda.synchronize()
    runtime.set_profile_buffer(None)
    t = buf.cpu().numpy()
    total = t[n_ops * 8 + 1] - t[n_ops * 8]
    print(f"batch={B} guided program: {n_ops} ops ({prog.meta['n_den']} denoiser), traj_bytes={prog.traj_floats * 4}, kernel cycles (wg0) = {total}")
    print(f"{'op':>3} {'kind':>5} {'cout':>4} {'L':>3} {'mode':>4} {'nt':>2} {'ks':>2} {'fl':>3} {'kloop':>7} {'sync':>6} {'epi':>6} {'total':>7}")
    tk = ts = te = 0
    for i, op in enumerate(prog.ops):
        s0, s1, s2, s3, k4, k5, k6, k7 = t[i * 8:i * 8 + 8]
        if int(op[P2.W2_KIND]) == P2.KIND2_HEAD:
            print(f"{i:3d}  head (stamps of the previous op's slot are not written)")
            continue
        k, s, e = s1 - s0, s2 - s1, s3 - s2
        tk, ts, te = tk + k, ts + s, te + e
        print(f"{i:3d} {'den' if i < prog.meta['n_den'] else 'clf':>5} {op[P2.W2_COUT]:4d} {op[P2.W2_LOUT]:3d} {'4x4' if op[P2.W2_MODE] else '16':>4} {op[P2.W2_NT]:2d} "
              f"{op[P2.W2_KSPLIT]:2d} {op[P2.W2_FLAGS]:3d} {k:7d} {s:6d} {e:6d} {s3 - s0:7d} | decode {k4 - s0:5d} operands {k5 - k4:5d} mfma {k6 - k5:6d} "
              f"stage {k7 - k6:5d} prefetch {s1 - k7:5d}")
    print(f"totals: kloop={tk} sync={ts} epilogue={te}  sum={tk + ts + te}")


if __name__ == "__main__":
    main()
