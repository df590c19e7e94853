"""Batch-sharded sampling across the GPUs of one node (one process per GPU, ``torch.distributed``).

The reference has no distributed code; sampling shards trivially because trajectories are independent and the
weights (<= 276 MB) are replicated.  There is NO collective on the data path: rank r denoises rows
``[r*B/N, (r+1)*B/N)`` of the request.  The only optional exchange is one ``all_gather`` of the finished (B/N, ...) fp32
shards (RCCL over xGMI on GPUs, gloo on CPU) when every rank needs the global result, e.g. for candidate arg-max.

Noise is drawn for the GLOBAL batch from a seeded CPU generator and sliced per rank, so the result is independent of the
number of ranks (the property the world_size-2 gloo test checks).
"""
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous, near-equal split of n rows: the first n % world ranks get one extra row."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_noise(shape, n_draws: int, seed: int) -> List[torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(shape, generator=g) for _ in range(n_draws)]


def _all_gather_rows(local: torch.Tensor, n: int, world: int) -> torch.Tensor:
    """Concatenate the (rows_r, ...) shards of an n-row tensor over the ranks: ONE all-gather (RCCL on GPUs) -- straight into
    the result when the split is even, through a padded buffer when it is ragged."""
    local = local.contiguous()
    if n % world == 0:
        out = torch.empty((n, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    biggest = max(b - a for a, b in sizes)
    pad = torch.zeros((biggest, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: b - a] for p, (a, b) in zip(parts, sizes)], dim=0)


# keyword arguments of ``sample()`` that carry one row per sample (reference diffusionsde.py:401-412): they are cut to the rank's
# rows together with `prior`; everything else (schedules, weights, flags, per-dimension bounds) is passed through
_PER_SAMPLE_KWARGS = ("condition_cfg", "condition_cg", "mask_cfg", "mask_cg", "warm_start_reference")


def sharded_sample(agent, prior: torch.Tensor, *, gather: bool = True, seed: Optional[int] = None,
                   return_logp: bool = False, **sample_kwargs):
    """Run ``agent.sample`` on this rank's slice of `prior` (global tensor, identical on every rank).

    Returns the global (B, ...) result on every rank if ``gather`` else the local shard.  Every per-sample tensor argument
    (conditions, masks, the warm-start reference, a recorded ``noise=[...]`` list) is sliced to the rank's rows.  ``seed`` draws
    the global noise list on the CPU and slices it (rank-count independent results); without it and without ``noise`` each rank
    uses its own RNG.  ``return_logp``: also return the classifier score ``log["log_p"]`` (B, 1) -- gathered like the samples --
    so that candidate selection (Diffuser: arg-max over the candidates of an environment) happens on the global batch after the
    one exchange.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    n = prior.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    kw = dict(sample_kwargs)
    kw["n_samples"] = hi - lo
    for name in _PER_SAMPLE_KWARGS:
        v = kw.get(name)
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n:
            kw[name] = v[lo:hi]
    if kw.get("noise") is not None:
        kw["noise"] = [z[lo:hi] if z.shape[0] == n else z for z in kw["noise"]]
    elif seed is not None:
        steps = kw.get("sample_steps")
        if steps is None:
            # the default of the agent's own sample() signature: 5 for the SDE solvers, None = "every diffusion step" only for
            # the legacy DDPM class (drawing diffusion_steps + 1 = 1001 global tensors for a 5-step call costs GBs of host memory)
            import inspect
            par = inspect.signature(agent.sample).parameters.get("sample_steps")
            steps = par.default if par is not None and par.default is not inspect.Parameter.empty else None
            if steps is None:
                steps = getattr(agent, "diffusion_steps", 5)
        n_draws = steps + (kw.get("diffusion_x_sampling_steps") or 0) + 1
        ref = sample_kwargs.get("warm_start_reference")
        shape = tuple(ref.shape) if isinstance(ref, torch.Tensor) else tuple(prior.shape)   # the draws follow the tensor they perturb
        kw["noise"] = [z[lo:hi] for z in global_noise(shape, n_draws, seed)]
    x, log = agent.sample(prior[lo:hi], **kw)
    logp = log.get("log_p") if return_logp else None
    if return_logp and logp is None:
        raise ValueError("return_logp needs a solver with a classifier (log['log_p'] is absent)")
    if gather and world > 1:
        x = _all_gather_rows(x, n, world)
        if logp is not None:
            logp = _all_gather_rows(logp, n, world)
    return (x, logp) if return_logp else x
