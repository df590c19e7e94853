"""Where does the HOST time of one steady-state sample() call go?  cProfile over 300 calls of config 2 at a small batch (GPU box).
Usage: python tools/host_profile.py [batch]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda", 0)
    agent, net = bench.build_agent(dev)
    prior, z0 = bench.make_inputs(dev, 0, batch)
    kw = dict(solver="ddim", n_samples=batch, sample_steps=20, temperature=0.5)
    for _ in range(5):
        agent.sample(prior, **kw)
    torch.cuda.synchronize()
    os.environ["CDX_UNET2_SPLIT_SYNC"] = "0"
    t0 = time.perf_counter()
    for _ in range(300):
        agent.sample(prior, **kw)
    host = (time.perf_counter() - t0) / 300
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300
    print(f"B={batch}: host time per call (no sync) {1e3 * host:.3f} ms, wall per call {1e3 * wall:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        agent.sample(prior, **kw)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
