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
    # Diffuser tail: classifier scores travel with the samples, candidate arg-max happens on the gathered batch
    scorer, _ = cases.build(lib, "janner_cfg2_diffuser_logp")
    c = cases.CASES["janner_cfg2_diffuser_logp"]
    prior2 = torch.zeros(7, c["horizon"], c["net"][1]["in_dim"])
    xs, logp = sharded_sample(scorer, prior2, gather=True, seed=3, return_logp=True, solver="ddim", sample_steps=3, temperature=0.5)
    # per-sample tensor kwargs are sliced with the prior: CFG condition (even 4 + 4 split -> all_gather_into_tensor) and a
    # recorded noise list
    cond_agent, _ = cases.build(lib, "janner_tiny_cond_w2")
    cc = cases.CASES["janner_tiny_cond_w2"]
    inp = cases.make_inputs("janner_tiny_cond_w2")
    reps = -(-8 // inp["prior"].shape[0])
    prior3 = torch.from_numpy(inp["prior"]).repeat(reps, 1, 1)[:8]
    cond3 = torch.from_numpy(inp["cond"]).repeat(reps, *([1] * (inp["cond"].ndim - 1)))[:8] * torch.linspace(0.5, 1.5, 8).view(-1, *([1] * (inp["cond"].ndim - 1)))
    g = torch.Generator().manual_seed(5)
    zs = [torch.randn(prior3.shape, generator=g) for _ in range(cc["sample"]["sample_steps"] + 1)]
    kw3 = {k: v for k, v in cases.sample_kwargs("janner_tiny_cond_w2", inp).items() if k not in ("n_samples", "condition_cfg")}
    xc = sharded_sample(cond_agent, prior3, gather=True, noise=zs, condition_cfg=cond3, **kw3)
    if rank == 0:
        torch.save({"x": x, "xs": xs, "logp": logp, "xc": xc, "prior3": prior3, "cond3": cond3, "zs": zs}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process(tmp_path):
    out = str(tmp_path / "x.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    x2 = got["x"]
    sys.path.insert(0, ROOT)
    from cleandiffuser_amd.distributed import sharded_sample, shard_bounds
    from oracle import cases
    agent, _ = cases.build(cases.lib_namespace("amd"), "janner_tiny_disc_ddpm")
    x1 = sharded_sample(agent, torch.zeros(5, 8, 6), gather=True, seed=11, solver="ddpm", sample_steps=5,
                        temperature=0.8)
    assert x2.shape == (5, 8, 6)
    # ATen CPU kernels block differently for batch 5 vs 3+2, so agreement is to rounding, not bitwise
    assert torch.allclose(x1, x2, rtol=1e-4, atol=1e-4)
    scorer, _ = cases.build(cases.lib_namespace("amd"), "janner_cfg2_diffuser_logp")
    c = cases.CASES["janner_cfg2_diffuser_logp"]
    xs1, logp1 = sharded_sample(scorer, torch.zeros(7, c["horizon"], c["net"][1]["in_dim"]), gather=True, seed=3, return_logp=True,
                                solver="ddim", sample_steps=3, temperature=0.5)
    assert got["logp"].shape == (7, 1) and torch.allclose(got["xs"], xs1, rtol=1e-4, atol=1e-4)
    assert torch.allclose(got["logp"], logp1, rtol=1e-4, atol=1e-4) and int(got["logp"].argmax()) == int(logp1.argmax())
    cond_agent, _ = cases.build(cases.lib_namespace("amd"), "janner_tiny_cond_w2")
    inp = cases.make_inputs("janner_tiny_cond_w2")
    kw3 = {k: v for k, v in cases.sample_kwargs("janner_tiny_cond_w2", inp).items() if k not in ("n_samples", "condition_cfg")}
    xc1 = sharded_sample(cond_agent, got["prior3"], gather=True, noise=got["zs"], condition_cfg=got["cond3"], **kw3)
    assert got["xc"].shape == xc1.shape and torch.allclose(got["xc"], xc1, rtol=1e-4, atol=1e-4), "per-sample kwargs must be sliced"
    assert [shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [shard_bounds(0, r, 2) for r in range(2)] == [(0, 0), (0, 0)]


def _worker_small(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cleandiffuser_amd.distributed import sharded_sample
    from oracle import cases
    lib = cases.lib_namespace("amd")
    agent, _ = cases.build(lib, "janner_tiny_disc_ddpm")
    outs = {n: sharded_sample(agent, torch.zeros(n, 8, 6), gather=True, seed=11, solver="ddpm", sample_steps=5, temperature=0.8) for n in (1, 4)}
    scorer, _ = cases.build(lib, "janner_cfg2_diffuser_logp")
    c = cases.CASES["janner_cfg2_diffuser_logp"]
    xs, logp = sharded_sample(scorer, torch.zeros(2, c["horizon"], c["net"][1]["in_dim"]), gather=True, seed=3, return_logp=True,
                              solver="ddim", sample_steps=3, temperature=0.5)
    if rank == world - 1:                                     # the rank whose shard was EMPTY for n = 1 and n = 2 reports
        torch.save({"outs": outs, "xs": xs, "logp": logp}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_three_ranks_with_empty_shards(tmp_path):
    """Fewer rows than ranks (the tail of a strong-scaling run): ranks with an empty shard still take part in the one all-gather and
    every rank ends with the global result."""
    out = str(tmp_path / "y.pt")
    mp.spawn(_worker_small, args=(3, _free_port(), out), nprocs=3, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from cleandiffuser_amd.distributed import sharded_sample, shard_bounds
    from oracle import cases
    assert [shard_bounds(1, r, 3) for r in range(3)] == [(0, 1), (1, 1), (1, 1)]
    agent, _ = cases.build(cases.lib_namespace("amd"), "janner_tiny_disc_ddpm")
    for n in (1, 4):
        want = sharded_sample(agent, torch.zeros(n, 8, 6), gather=True, seed=11, solver="ddpm", sample_steps=5, temperature=0.8)
        assert got["outs"][n].shape == (n, 8, 6) and torch.allclose(got["outs"][n], want, rtol=1e-4, atol=1e-4)
    scorer, _ = cases.build(cases.lib_namespace("amd"), "janner_cfg2_diffuser_logp")
    c = cases.CASES["janner_cfg2_diffuser_logp"]
    xs1, logp1 = sharded_sample(scorer, torch.zeros(2, c["horizon"], c["net"][1]["in_dim"]), gather=True, seed=3, return_logp=True,
                                solver="ddim", sample_steps=3, temperature=0.5)
    assert got["logp"].shape == (2, 1) and torch.allclose(got["xs"], xs1, rtol=1e-4, atol=1e-4) and torch.allclose(got["logp"], logp1, rtol=1e-4, atol=1e-4)


def _worker_eight(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cleandiffuser_amd.distributed import sharded_sample, shard_bounds
    from oracle import cases
    scorer, _ = cases.build(cases.lib_namespace("amd"), "janner_cfg2_diffuser_logp")
    g = torch.Generator().manual_seed(8)
    prior = torch.zeros(256, 32, 23)
    prior[:, 0, :17] = torch.randn(256, 17, generator=g)
    assert shard_bounds(256, rank, world) == (32 * rank, 32 * rank + 32)
    xs, logp = sharded_sample(scorer, prior, gather=True, seed=4, return_logp=True, solver="ddim", sample_steps=2, temperature=0.5)
    if rank == world - 1:
        torch.save({"xs": xs, "logp": logp, "prior": prior}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_the_metric_batch(tmp_path):
    """The 8-GPU shape of the headline metric (VERDICT r3 'next' #7): global B = 256 of config 2 with its classifier, 32 trajectories
    per rank, ONE all-gather of (256, 32, 23) plus one of log_p (256, 1); every rank ends with the global batch, candidate selection
    (arg-max of log_p over the gathered batch, and per environment of 64 candidates) equals the single-process result index for index."""
    out = str(tmp_path / "z.pt")
    mp.spawn(_worker_eight, args=(8, _free_port(), out), nprocs=8, join=True)
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from cleandiffuser_amd.distributed import sharded_sample
    from oracle import cases
    scorer, _ = cases.build(cases.lib_namespace("amd"), "janner_cfg2_diffuser_logp")
    xs1, logp1 = sharded_sample(scorer, got["prior"], gather=True, seed=4, return_logp=True, solver="ddim", sample_steps=2, temperature=0.5)
    assert got["xs"].shape == (256, 32, 23) and got["logp"].shape == (256, 1)
    assert torch.allclose(got["xs"], xs1, rtol=1e-4, atol=1e-4) and torch.allclose(got["logp"], logp1, rtol=1e-4, atol=1e-4)
    assert int(got["logp"].argmax()) == int(logp1.argmax())
    assert torch.equal(got["logp"].view(4, 64).argmax(1), logp1.view(4, 64).argmax(1))      # 4 environments x 64 candidate plans
