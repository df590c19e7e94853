#!/usr/bin/env python
"""bench.py -- denoised trajectories/sec on BASELINE config 2 (the north-star metric).

One "step" = one complete ``DiscreteDiffusionSDE.sample()`` call: JannerUNet1d (in 23, model 32, dim_mult [1,2,2,2],
k=5), horizon 32, B=256 trajectories per GPU, 20-step DDIM, x-prediction, fix-mask on the first observation,
temperature 0.5 -- synthetic random-init weights and synthetic inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (driver, N > 1)

Multi-GPU: trajectories are independent, so each rank denoises its own 256 (weak scaling, no data-path
collective); the only collectives are the timing barrier and the MAX reduction of the elapsed time.

Prints ONE JSON line (rank 0).  ``roofline`` prices the single fused kernel against the fp32-MFMA peak using the
algorithmic FLOPs of the reference modules (786.6 MFLOP per trajectory = 39.33 MFLOP x 20 forwards) and
the kernel's mean duration from HIP events on the launch stream.  ``cpu_baseline`` times the CPU oracle port
(oracle/torch_port.py, the same ATen ops the reference runs) on this host's cores -- reported baseline only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, HORIZON, DIM, SAMPLE_STEPS = int(os.environ.get("BENCH_BATCH", "256")), 32, 23, 20
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak


def build_agent(device):
    from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    from cleandiffuser_amd.utils import load_synth
    net = load_synth(JannerUNet1d(DIM, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), seed=0)
    fix_mask = torch.zeros(HORIZON, DIM)
    fix_mask[0, :17] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix_mask, diffusion_steps=SAMPLE_STEPS, predict_noise=False,
                                 device=device)
    agent.eval()
    return agent, net


def make_inputs(device, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    prior = torch.zeros(BATCH, HORIZON, DIM)
    prior[:, 0, :17] = torch.randn(BATCH, 17, generator=g)
    z0 = torch.randn(BATCH, HORIZON, DIM, generator=g)
    return prior.to(device), z0.to(device)


def cpu_baseline(net, budget_s=12.0):
    """Time the CPU oracle on a bounded sample of the same workload: whole sample() calls at B=256 until
    ~budget_s of CPU work has been done (>= 2 calls)."""
    from oracle import torch_port
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    fwd = torch_port.make_forward(sd, dict(emb_dim=32, kernel_size=5, dim_mult=[1, 2, 2, 2]))
    prior, z0 = make_inputs("cpu", 0)
    fm = torch.zeros(1, HORIZON, DIM)
    fm[0, 0, :17] = 1.0
    avail = torch.get_num_threads()           # torch's own default = the cores this process may use

    def call():
        with torch.no_grad():
            return torch_port.vp_sample(fwd, prior, [z0], solver="ddim", sample_steps=SAMPLE_STEPS, discrete=True,
                                        diffusion_steps=SAMPLE_STEPS, temperature=0.5, predict_noise=False,
                                        fix_mask=fm)
    # the reference path is small-op bound: all cores of a big host oversubscribe it, so probe a few thread counts
    # (one timed call each) and run the bounded sample at the fastest -- `cores` reports what was actually used
    best, cores = None, avail
    for th in sorted({t for t in (8, 16, 32, 64, avail) if t <= avail}):
        torch.set_num_threads(th)
        call()                                 # warm-up (thread pool, oneDNN primitives)
        t0 = time.perf_counter()
        call()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
    torch.set_num_threads(cores)
    call()
    t0, n = time.perf_counter(), 0
    while n < 2 or time.perf_counter() - t0 < budget_s:
        call()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": BATCH * n / dt, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"{n} full sample() calls of B={BATCH} (20-step DDIM) through oracle/torch_port.py, "
                      f"{dt:.1f}s wall, torch {torch.__version__} CPU, {cores} of {avail} threads (fastest of a probe)"}


def cpu_baseline_subprocess(timeout_s=180):
    """Run the CPU leg in a child with a hard wall-clock bound so the GPU line is never held hostage."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                           text=True, timeout=timeout_s)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 -- report, never fail the bench on the baseline leg
        return {"value": None, "unit": "trajectories/s", "cores": None, "kind": "port",
                "sample": f"cpu baseline leg failed or exceeded {timeout_s}s: {type(e).__name__}"}


def recorded_traffic():
    """HBM-side bytes per launch of the fused kernel from the committed PMC pass (rocprofv3 --pmc cannot run inside this
    process); null when the record is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return {"traffic": rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"], "traffic_unit": "bytes/launch",
                "traffic_source": "profiles/r01_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass, gfx950-corrected)"}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:               # child process of the N=1 run: CPU oracle timing only
        _, net = build_agent("cpu")
        print(json.dumps(cpu_baseline(net)), flush=True)
        return

    import faulthandler
    faulthandler.dump_traceback_later(240, exit=False)     # if anything wedges, say where (stderr)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from cleandiffuser_amd.engine import runtime
    runtime.load_library()
    agent, net = build_agent(device)
    prior, z0 = make_inputs(device, rank)
    kw = dict(solver="ddim", n_samples=BATCH, sample_steps=SAMPLE_STEPS, temperature=0.5)

    def step():
        x, _ = agent.sample(prior, noise=[z0], **kw)
        return x

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    runtime.enable_launch_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = runtime.drain_launch_timing()
    runtime.enable_launch_timing(False)
    assert torch.isfinite(x).all()

    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        prog = runtime.compiled_program(agent.model_ema["diffusion"], HORIZON).prog
        flops_per_traj = 2.0 * prog.macs_per_forward * SAMPLE_STEPS
        k_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        achieved = flops_per_traj * BATCH / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        out = {
            "metric": "denoised trajectories/sec @ (B=256,H=32,D=23) 20-step DDIM",
            "value": BATCH * world * args.steps / elapsed,
            "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: JannerUNet1d Diffuser H=32 D=23, 20-step DDIM, "
                                   "B=256 trajectories per GPU, whole DiscreteDiffusionSDE.sample() call",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "horizon": HORIZON, "dim": DIM,
                       "sample_steps": SAMPLE_STEPS, "parallelism": f"batch-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, **recorded_traffic(),
                         "kernel": "cdx_unet2_kernel" if os.environ.get("CDX_UNET2", "1") != "0" else "cdx_unet1d_kernel", "kernel_ms": k_ms, "launches_timed": len(kernel_ms),
                         "flops_per_launch": flops_per_traj * BATCH},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess()
        print(json.dumps(out), flush=True)
    faulthandler.cancel_dump_traceback_later()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
