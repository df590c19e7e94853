"""Kernel census of ONE update() step (eager library nodes, so that the profiler can name the kernels): launches and device time per
kernel family.  Usage (GPU box): python tools/update_census.py [cfg2|cfg3|cfg4|cfg5|chitf]"""
import collections
import os
import re
import sys

os.environ["CDX_TRAIN_GRAPH"] = "0"
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from update_bench import build  # noqa: E402


def family(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    if "at::native" in name:
        m = re.search(r"(CUDAFunctor_add|FillFunctor|direct_copy|flip_kernel|reduce_kernel|CatArray|mish_backward|mish_kernel|BinaryFunctor|AUnaryFunctor|"
                      r"BUnaryFunctor|index_|bernoulli|masked_fill|where|mul|neg|sum)", name)
        return "aten:" + (m.group(1) if m else name[:60])
    return re.sub(r"^\(anonymous namespace\)::", "", re.sub(r"(?<!^)\(.*", "", name))[:70]


def main():
    name = (sys.argv[1:] or ["cfg2"])[0]
    steps = 5
    agent, x0, cond, what = build(name)
    call = (lambda: agent.update(x0, cond)) if cond is not None else (lambda: agent.update(x0))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            call()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.key_averages():
        if e.device_time_total > 0 or "Memcpy" in e.key or "Memset" in e.key:
            a = agg[family(e.key)]
            a[0] += e.count
            a[1] += e.device_time_total
    tot_n, tot_t = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
    print(f"{what}: {tot_n / steps:.0f} launches, {tot_t / steps / 1e3:.3f} ms of device time per update()")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:72s} {n / steps:7.1f} launches {t / steps:9.1f} us {100 * t / tot_t:5.1f} %")


if __name__ == "__main__":
    main()
