"""Which ATen ops does one steady-state sample() call of the north-star config dispatch on the host side?  (GPU box)
Prints the op list of the 4th call; the fused path should show no compute ops besides the kernel launch itself."""
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.ops.append(str(func))
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda", 0)
    agent, net = bench.build_agent(dev)
    prior, z0 = bench.make_inputs(dev, 0)
    kw = dict(solver="ddim", n_samples=bench.BATCH, sample_steps=20, temperature=0.5)
    for _ in range(3):
        agent.sample(prior, noise=[z0], **kw)
    with Log() as log:
        agent.sample(prior, noise=[z0], **kw)
    torch.cuda.synchronize()
    print(len(log.ops), "ATen ops in one steady-state sample() call:")
    for o in log.ops:
        print("  ", o)


if __name__ == "__main__":
    main()
