"""Generate tests/golden/extra_*.npz by running the REAL reference (imported from /root/reference) on oracle/extra_cases.py.
Build container only: ``python -m oracle.gen_golden_extra [name ...]``.  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import os
import sys

import numpy as np

from . import extra_cases


def main(out_dir="tests/golden", only=None):
    os.makedirs(out_dir, exist_ok=True)
    for name in extra_cases.SCENARIOS:
        if only and name not in only:
            continue
        out = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in extra_cases.run(name, "reference").items()
               if not k.startswith("_")}
        assert all(np.isfinite(v).all() for v in out.values()), name
        np.savez_compressed(os.path.join(out_dir, f"extra_{name}.npz"), **out)
        print(f"{name:24s} " + " ".join(f"{k}{v.shape} |max|={np.abs(v).max():.3f}" for k, v in out.items()))


if __name__ == "__main__":
    main(only=sys.argv[1:] or None)
