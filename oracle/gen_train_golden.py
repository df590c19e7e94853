"""Generate tests/golden/train_*.npz by running the REAL reference's loss() / update() on oracle/train_cases.py.
Build container only: ``python -m oracle.gen_train_golden [name ...]``.  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import os
import sys

import numpy as np

from . import train_cases


def main(out_dir="tests/golden", only=None):
    os.makedirs(out_dir, exist_ok=True)
    for name in train_cases.SCENARIOS:
        if only and name not in only:
            continue
        out = train_cases.run(name, "reference")
        assert all(np.isfinite(v).all() for v in out.values()), name
        np.savez_compressed(os.path.join(out_dir, f"train_{name}.npz"), **out)
        print(f"{name:22s} loss={out['loss'][0]:.6f} upd={out['upd_loss']}" + (f" gn={out['grad_norm']}" if "grad_norm" in out else ""))


if __name__ == "__main__":
    main(only=sys.argv[1:] or None)
