"""N > 1 path on CPU: two gloo ranks shard a request, sample locally, all-gather -- and must reproduce the
single-process result exactly (global noise is sliced, so the answer cannot depend on the rank count)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cleandiffuser_amd.distributed import sharded_sample
    from oracle import cases
    lib = cases.lib_namespace("amd")
    agent, _ = cases.build(lib, "janner_tiny_disc_ddpm")
    prior = torch.zeros(5, 8, 6)                              # 5 rows over 2 ranks: ragged 3 + 2 split
    x = sharded_sample(agent, prior, gather=True, seed=11, solver="ddpm", sample_steps=5, temperature=0.8)
    if rank == 0:
        torch.save(x, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process(tmp_path):
    out = str(tmp_path / "x.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    x2 = torch.load(out)
    sys.path.insert(0, ROOT)
    from cleandiffuser_amd.distributed import sharded_sample, shard_bounds
    from oracle import cases
    agent, _ = cases.build(cases.lib_namespace("amd"), "janner_tiny_disc_ddpm")
    x1 = sharded_sample(agent, torch.zeros(5, 8, 6), gather=True, seed=11, solver="ddpm", sample_steps=5,
                        temperature=0.8)
    assert x2.shape == (5, 8, 6)
    # ATen CPU kernels block differently for batch 5 vs 3+2, so agreement is to rounding, not bitwise
    assert torch.allclose(x1, x2, rtol=1e-4, atol=1e-4)
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(0, r, 2) for r in range(2)] == [(0, 0), (0, 0)]
