"""BUILD CONTAINER ONLY: time the REAL reference (/root/reference, imported through oracle/ref_import.py) AND the oracle port
(oracle/torch_port.py -- what bench.py's ``cpu_baseline`` leg has to run on the GPU box, where /root/reference does not exist) on
BASELINE config 2 -- DiscreteDiffusionSDE.sample(), JannerUNet1d H=32 D=23, 20-step DDIM, B=256 -- back to back in this container:
same weights, same inputs, same thread count, calls interleaved (port, reference, port, ...), best-of timing per implementation.
The record, profiles/r<NN>_reference_cpu.json (NN = argv[1], default 05), carries both rates and their ratio ``port_over_reference``;
bench.py prints that ratio next to the port figure it measures, so the GPU-box number is traceable to the reference
(tests/test_bench_contract.py re-measures the ratio against the record).
Usage: python tools/measure_reference_cpu.py [round]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases, torch_port  # noqa: E402

H, D, B, STEPS = 32, 23, 256, 20


def build_pair():
    """(reference call, port call) on the same weights and inputs."""
    ref = cases.lib_namespace("reference")
    torch.manual_seed(0)
    net = ref.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5)
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    agent = ref.DiscreteDiffusionSDE(net, None, fix_mask=fix, diffusion_steps=STEPS, predict_noise=False, device="cpu")
    agent.eval()
    prior = torch.zeros(B, H, D)
    prior[:, 0, :17] = torch.randn(B, 17)
    z0 = torch.randn(B, H, D)
    fwd = torch_port.make_forward({k: v.detach() for k, v in net.state_dict().items()}, dict(emb_dim=32, kernel_size=5, dim_mult=[1, 2, 2, 2]))

    def ref_call():
        with torch.no_grad():
            return agent.sample(prior, solver="ddim", n_samples=B, sample_steps=STEPS, temperature=0.5)[0]

    def port_call():
        with torch.no_grad():
            return torch_port.vp_sample(fwd, prior, [z0], solver="ddim", sample_steps=STEPS, discrete=True, diffusion_steps=STEPS,
                                        temperature=0.5, predict_noise=False, fix_mask=fix[None])
    return ref_call, port_call


def interleaved(ref_call, port_call, threads, rounds):
    """Best-of-`rounds` seconds per call of each implementation, calls interleaved so that both see the same machine state."""
    torch.set_num_threads(threads)
    ref_call(), port_call()                                   # warm-up (thread pool, oneDNN primitives)
    best = {"reference": float("inf"), "port": float("inf")}
    for _ in range(rounds):
        for name, fn in (("port", port_call), ("reference", ref_call)):
            t0 = time.perf_counter()
            fn()
            best[name] = min(best[name], time.perf_counter() - t0)
    return best


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "05"
    ref_call, port_call = build_pair()
    avail = torch.get_num_threads()
    legs = {}
    for tag, th, rounds in (("all_threads", avail, 6), ("one_thread", 1, 2)):
        best = interleaved(ref_call, port_call, th, rounds)
        legs[tag] = {"threads": th, "calls_each": rounds, "timing": "best of the interleaved calls",
                     "reference": {"value": B / best["reference"], "unit": "trajectories/s"},
                     "port": {"value": B / best["port"], "unit": "trajectories/s"},
                     "port_over_reference": best["reference"] / best["port"]}
    model = "unknown"
    with open("/proc/cpuinfo") as f:
        for line in f:
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    rec = {"what": "the real CleanDiffuser reference (imported from /root/reference) and the oracle port (oracle/torch_port.py), BASELINE "
                   "configs[1], B=256, 20-step DDIM, CPU, same weights / inputs / threads, interleaved calls",
           "where": "build container (not the GPU box)", "cpu_model": model, "torch": torch.__version__,
           "all_threads": {"value": legs["all_threads"]["reference"]["value"], "unit": "trajectories/s", "threads": avail,
                           "calls": legs["all_threads"]["calls_each"]},
           "one_thread": {"value": legs["one_thread"]["reference"]["value"], "unit": "trajectories/s", "threads": 1,
                          "calls": legs["one_thread"]["calls_each"]},
           "port_vs_reference": legs, "port_over_reference": legs["all_threads"]["port_over_reference"],
           "script": "tools/measure_reference_cpu.py", "round": int(rnd), "measured_on": time.strftime("%Y-%m-%d")}
    with open(os.path.join(ROOT, "profiles", f"r{rnd}_reference_cpu.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
