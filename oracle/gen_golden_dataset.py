"""Write tests/golden/dataset_*.npz from the IMPORTED reference datasets (build container only):
``python -m oracle.gen_golden_dataset``.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Per sequence scenario: N_ITEMS collated items (obs / act / rew / val) at recorded indices, the item table, a checksum of every
constructor array (float64 sum, sum of squares) and the normaliser; per transition scenario: the same for the five transition tensors."""
import copy
import os

import numpy as np
import torch
from torch.utils.data import default_collate

from . import dataset_cases as dc
from .ref_import import import_reference


def _sums(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), (a * a).sum()])


def main(out_dir="tests/golden"):
    import_reference()
    from cleandiffuser.dataset.d4rl_mujoco_dataset import D4RLMuJoCoDataset, D4RLMuJoCoTDDataset
    for name, (skw, dkw) in dc.SCENARIOS.items():
        ds = D4RLMuJoCoDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
        idx = dc.item_indices(len(ds), skw["seed"])
        b = default_collate([ds[int(i)] for i in idx])
        out = dict(idx=idx, obs=b["obs"]["state"].numpy(), act=b["act"].numpy(), rew=b["rew"].numpy(), val=b["val"].numpy(),
                   indices=np.array(ds.indices, dtype=np.int64), path_lengths=ds.path_lengths,
                   mean=ds.get_normalizer().mean, std=ds.get_normalizer().std,
                   **{f"sum_{k}": _sums(getattr(ds, k)) for k in ("seq_obs", "seq_act", "seq_rew", "seq_val")})
        np.savez_compressed(os.path.join(out_dir, f"dataset_{name}.npz"), **out)
        print(f"{name:24s} items={len(ds)} paths={len(ds.path_lengths)}")
    for name, (skw, dkw) in dc.TD_SCENARIOS.items():
        ds = D4RLMuJoCoTDDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
        idx = dc.item_indices(len(ds), skw["seed"])
        b = default_collate([ds[int(i)] for i in idx])
        out = dict(idx=idx, obs=b["obs"]["state"].numpy(), next_obs=b["next_obs"]["state"].numpy(), act=b["act"].numpy(),
                   rew=b["rew"].numpy(), tml=b["tml"].numpy(),
                   **{f"sum_{k}": _sums(getattr(ds, k).numpy()) for k in ("obs", "next_obs", "act", "rew", "tml")})
        np.savez_compressed(os.path.join(out_dir, f"dataset_{name}.npz"), **out)
        print(f"{name:24s} items={len(ds)}")


if __name__ == "__main__":
    main()
