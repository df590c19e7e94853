"""BUILD CONTAINER ONLY: time the REAL reference (/root/reference, imported through oracle/ref_import.py) on BASELINE config 2
-- DiscreteDiffusionSDE.sample(), JannerUNet1d H=32 D=23, 20-step DDIM, B=256 -- on this container's CPU, at torch's default
thread count and at one thread, and record it as profiles/r<NN>_reference_cpu.json (NN = the round, argv[1]; default 04) with the date of the measurement.  bench.py carries the record along as a side
figure next to its own cpu_baseline (which has to run on the GPU box, where /root/reference does not exist).
Usage: python tools/measure_reference_cpu.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "04"
    ref = cases.lib_namespace("reference")
    torch.manual_seed(0)
    net = ref.JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5)
    fix = torch.zeros(32, 23)
    fix[0, :17] = 1.0
    agent = ref.DiscreteDiffusionSDE(net, None, fix_mask=fix, diffusion_steps=20, predict_noise=False, device="cpu")
    agent.eval()
    prior = torch.zeros(256, 32, 23)
    prior[:, 0, :17] = torch.randn(256, 17)

    def call():
        with torch.no_grad():
            return agent.sample(prior, solver="ddim", n_samples=256, sample_steps=20, temperature=0.5)[0]

    def rate(threads, budget):
        torch.set_num_threads(threads)
        call()
        t0, n = time.perf_counter(), 0
        while n < 2 or time.perf_counter() - t0 < budget:
            call()
            n += 1
        return 256 * n / (time.perf_counter() - t0), n

    avail = torch.get_num_threads()
    v_all, n_all = rate(avail, 15.0)
    v_one, n_one = rate(1, 10.0)
    model = "unknown"
    with open("/proc/cpuinfo") as f:
        for line in f:
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    rec = {"what": "the real CleanDiffuser reference (imported from /root/reference), BASELINE configs[1], B=256, 20-step DDIM, CPU",
           "where": "build container (not the GPU box)", "cpu_model": model, "torch": torch.__version__,
           "all_threads": {"value": v_all, "unit": "trajectories/s", "threads": avail, "calls": n_all},
           "one_thread": {"value": v_one, "unit": "trajectories/s", "threads": 1, "calls": n_one},
           "script": "tools/measure_reference_cpu.py", "round": int(rnd), "measured_on": time.strftime("%Y-%m-%d")}
    with open(os.path.join(ROOT, "profiles", f"r{rnd}_reference_cpu.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
