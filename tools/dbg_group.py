"""Debug aid: one grouped launch of config 2 against the ordinary program on the same request (max |d|, error words)."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleandiffuser_amd.engine import runtime2  # noqa: E402

dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
agent, net = bench.build_agent(dev)
prior, z0 = bench.make_inputs(dev, 0, batch)
kw = dict(solver="ddim", n_samples=batch, sample_steps=int(os.environ.get("STEPS", "20")), temperature=0.5)
os.environ["CDX_UNET2_GROUP"] = "0"; os.environ["CDX_UNET2_SPLIT"] = "0"
ref, _ = agent.sample(prior, noise=[z0], **kw)
del os.environ["CDX_UNET2_GROUP"]; del os.environ["CDX_UNET2_SPLIT"]
os.environ["CDX_UNET2_REPAIR"] = "0"
runtime2._group_ok[dev] = True; runtime2._split_ok[dev] = True       # skip the first-use check: look at the raw result
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    out, _ = agent.sample(prior, noise=[z0], **kw)
    torch.cuda.synchronize()
    for x in w:
        print("WARN", x.message)
ent = runtime2._split_errs.get(dev)
print("error words", None if ent is None else ent[1].tolist())
d = (out - ref).abs()
print("finite", bool(torch.isfinite(out).all()), "max|d|", float(torch.nan_to_num(d, nan=1e9).max()), "rows wrong", int((torch.nan_to_num(d, nan=1e9).flatten(1).max(1).values > 1e-3).sum()), "of", batch)
bad = (torch.nan_to_num(d, nan=1e9).flatten(1).max(1).values > 1e-3).nonzero().flatten().tolist()
print("bad rows (first 32)", bad[:32])
