"""Per-op cycle breakdown of the v2 fused kernel (workgroup 0, second forward) -- tuning aid.
Usage (GPU box): python tools/op_profile2.py [batch] [traj_per_wg] [n_waves] > gpurun_out/op_profile2.txt
                 python tools/op_profile2.py 256 group4    (the grouped program, k = 4: workgroup 0 = member 0 of group 0; the `epi`
                                                            column of a grouped op includes its exchange)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleandiffuser_amd.engine import program2 as P2, runtime, runtime2  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    group = 0
    if len(sys.argv) > 2 and sys.argv[2].startswith("group"):
        group = int(sys.argv[2][5:])
        os.environ["CDX_UNET2_GROUP"] = str(group)
        os.environ["CDX_UNET2_GROUP_PROF"] = "1"
    elif len(sys.argv) > 2:
        os.environ["CDX_UNET2_T"] = sys.argv[2]
    if len(sys.argv) > 3:
        os.environ["CDX_UNET2_NW"] = sys.argv[3]
    bench.BATCH = batch
    dev = torch.device("cuda", 0)
    if os.environ.get("OP_PROFILE_FORCE_GROUP") == "1":     # diagnostic builds whose results are wrong on purpose: skip the first-use check
        runtime2._group_ok[dev] = True
    agent, net = bench.build_agent(dev)
    prior, z0 = bench.make_inputs(dev, 0)
    kw = dict(solver="ddim", n_samples=batch, sample_steps=20, temperature=0.5)
    for _ in range(3):
        agent.sample(prior, noise=[z0], **kw)
    comp, parts = runtime2.plan_for(agent.model_ema["diffusion"], 32, batch)
    prog, tpw = comp.prog, parts[0][2]
    if group:
        prog, tpw = runtime2.compiled_group2(agent.model_ema["diffusion"], 32, group).prog, 1
    n_ops = len(prog.ops)
    buf = torch.zeros(n_ops * 8 + 2, dtype=torch.int64, device=dev)
    runtime.set_profile_buffer(buf)
    agent.sample(prior, noise=[z0], **kw)
    torch.cuda.synchronize()
    runtime.set_profile_buffer(None)
    t = buf.cpu().numpy()
    total = t[n_ops * 8 + 1] - t[n_ops * 8]
    fwd = t[(n_ops - 1) * 8 + 3] - t[0]
    print(f"batch={batch} T={tpw} waves={prog.nw} group={group} kernel cycles (wg0) = {total}  traj_bytes={prog.traj_floats * 4}")
    print(f"first forward cycles = {fwd}  ({fwd * 20 / total:.2%} of kernel if all 20 equal)")
    print(f"{'op':>3} {'cout':>4} {'L':>3} {'mode':>4} {'nt':>2} {'ks':>2} {'nq/item':>7} {'kloop':>7} {'sync':>6} {'epi':>6} {'total':>7}")
    tk = ts = te = 0
    for i, op in enumerate(prog.ops):
        s0, s1, s2, s3, k4, k5, k6, k7 = t[i * 8:i * 8 + 8]
        nq = int(P2.op_item(prog.ops_buffer, op, 0)[P2.I2_NQ])
        k, s, e = s1 - s0, s2 - s1, s3 - s2
        tk, ts, te = tk + k, ts + s, te + e
        xg = int(op[P2.W2_XG])
        tag = "G" if xg & P2.XG_GOP else ("X" if xg & P2.XG_XCHG else " ")
        if os.environ.get("OP_PROFILE_FETCH"):      # a -DCDX2_PROF_FETCH=1 build: stamps 4-6 sit inside fetch_next
            print(f"{i:3d}{tag} kloop+stage {k7 - s0:6d} | item decode {k4 - k7:5d} ring loads {k5 - k4:5d} params {k6 - k5:5d} desc {s1 - k6:5d} | sync {s:5d} epi {e:6d}")
            continue
        if os.environ.get("OP_PROFILE_XCHG"):       # a -DCDX2_PROF_FETCH=2 build: stamps 4-6 sit around the exchange
            if xg & P2.XG_XCHG:
                print(f"{i:3d}{tag} kloop {k:6d} sync {s:5d} | epilogue {k4 - s2:5d} barrier+publish {k5 - k4:5d} collect {k6 - k5:5d} barrier {s3 - k6:5d} | epi column {e:6d}")
            continue
        print(f"{i:3d}{tag}{op[P2.W2_COUT]:4d} {op[P2.W2_LOUT]:3d} {'4x4' if op[P2.W2_MODE] else '16':>4} {op[P2.W2_NT]:2d} {op[P2.W2_KSPLIT]:2d} "
              f"{nq:7d} {k:7d} {s:6d} {e:6d} {s3 - s0:7d} | decode {k4 - s0:5d} operands {k5 - k4:5d} mfma {k6 - k5:6d} stage {k7 - k6:5d} prefetch {s1 - k7:5d}")
    print(f"totals: kloop={tk} sync={ts} epilogue={te}  sum={tk + ts + te}")


if __name__ == "__main__":
    main()
