"""Are torch's seeded CPU draws bit-identical on this host and on the build container?  (Round 5: the un-clipped config-4 scenarios amplify
an input difference of one ulp ~1000x; fixtures are made in the build container, the device test draws its inputs on the GPU box's host.)
Prints a digest of the draws the scenario baseline_cfg4_tied makes, and of torch's CPU capability."""
import hashlib
import sys

import numpy as np
import torch

g = torch.Generator().manual_seed(1000 + 4 + 3)
prior = torch.randn(3, 29, generator=g)
z = torch.randn(3, 64, 29, generator=g)
c = torch.randn(3, 1, generator=g)
for name, t in (("prior", prior), ("z", z), ("cond", c)):
    a = t.numpy()
    print(name, hashlib.sha256(a.tobytes()).hexdigest()[:16], repr(float(a.astype(np.float64).sum())))
print("cpu capability:", torch.backends.cpu.get_cpu_capability(), "| threads", torch.get_num_threads())
print("exp/sin/log probes:", repr(float(torch.exp(torch.tensor(0.7310585786)))), repr(float(torch.sin(torch.tensor(2.1234567)))),
      repr(float(torch.log(torch.tensor(0.3333333)))))
x = torch.linspace(0.1, 9.9, 4096)
print("vector exp/sin/log/silu/tanh digests:", *(hashlib.sha256(f(x).numpy().tobytes()).hexdigest()[:12]
                                                 for f in (torch.exp, torch.sin, torch.log, torch.nn.functional.silu, torch.tanh)))
