"""Per-op cycle breakdown of the batch-tiled MLP program (BASELINE config 1, PearceMlp): workgroup 0, first forward.
Usage (GPU box): python tools/op_profile_mlp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.engine import program as P, runtime  # noqa: E402
from tools import bench_configs as BC  # noqa: E402


def main():
    label, call, B, steps, net, horizon = BC.cfg1()
    for _ in range(2):
        call()
    prog = runtime.compiled_program(net, horizon).prog
    n_ops = len(prog.ops)
    buf = torch.zeros(n_ops * 8 + 2, dtype=torch.int64, device=BC.DEV)
    runtime.set_profile_buffer(buf)
    call()
    torch.cuda.synchronize()
    runtime.set_profile_buffer(None)
    t = buf.cpu().numpy()
    total = t[n_ops * 8 + 1] - t[n_ops * 8]
    fwd = t[(n_ops - 1) * 8 + 3] - t[0]
    print(f"{label}: kernel cycles (wg0) = {total}, first forward = {fwd} cycles ({fwd * steps / total:.1%} of the kernel if all "
          f"{steps} are equal), lds = {prog.lds_floats * 4} B, ops = {n_ops}")
    for i, op in enumerate(prog.ops):
        s0, s1, s2, s3, k1, k2, k3, k4 = t[i * 8:i * 8 + 8]
        kind = {P.OP_CONV: "conv", P.OP_LINEAR: "lin"}.get(op[P.W_KIND], "temb")
        if op[P.W_KIND] == P.OP_CONV:
            print(f"{i:3d} {kind:>5} cout {op[P.W_COUT]:4d} chunks {op[P.W_NCHUNKS]:4d} ksplit {op[P.W_KSPLIT]:2d} | kloop {s1 - s0:6d} "
                  f"sync {s2 - s1:6d} epilogue {s3 - s2:6d} total {s3 - s0:6d} | item {k1 - s0:5d} operands {k2 - k1:5d} "
                  f"mfma {k3 - k2:6d} tail {k4 - k3:5d}")
        else:
            print(f"{i:3d} {kind:>5} total {s3 - s0:6d}")
    gaps = [t[(i + 1) * 8] - t[i * 8 + 3] for i in range(n_ops - 1)]
    print("between-op gaps:", gaps)


if __name__ == "__main__":
    main()
