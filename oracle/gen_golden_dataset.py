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


def siblings(out_dir="tests/golden"):
    """Round 5: kitchen / antmaze / Decision-Veteran / multi-horizon classes (dataset_cases.SIBLING_*): the same records."""
    import importlib
    import_reference()
    for name, (cls, mod, skw, dkw) in dc.SIBLING_SCENARIOS.items():
        ds = getattr(importlib.import_module(f"cleandiffuser.dataset.{mod}"), cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
        idx = dc.item_indices(len(ds), skw["seed"])
        b = default_collate([ds[int(i)] for i in idx])
        out = dict(idx=idx, obs=b["obs"]["state"].numpy(), act=b["act"].numpy(), rew=b["rew"].numpy(), val=b["val"].numpy(),
                   indices=np.array(ds.indices, dtype=np.int64).reshape(-1, 3), mean=ds.get_normalizer().mean, std=ds.get_normalizer().std,
                   **{f"sum_{k}": _sums(getattr(ds, k)) for k in ("seq_obs", "seq_act", "seq_rew", "seq_val")})
        if "tml" in b:
            out["tml"] = b["tml"].numpy()
        if hasattr(ds, "paths"):
            out["paths"] = np.asarray(ds.paths, dtype=np.int64).reshape(-1, 2)
        if hasattr(ds, "tml_and_not_timeout"):
            out["tml_and_not_timeout"] = np.asarray(ds.tml_and_not_timeout, dtype=np.int64)
        np.savez_compressed(os.path.join(out_dir, f"dataset_{name}.npz"), **out)
        print(f"{name:24s} items={len(ds)} rows={np.asarray(ds.seq_obs).shape[:2]}")
    for name, (cls, mod, skw, dkw) in dc.SIBLING_TD_SCENARIOS.items():
        ds = getattr(importlib.import_module(f"cleandiffuser.dataset.{mod}"), cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
        idx = dc.item_indices(len(ds), skw["seed"])
        b = default_collate([ds[int(i)] for i in idx])
        out = dict(idx=idx, obs=b["obs"]["state"].numpy(), next_obs=b["next_obs"]["state"].numpy(), act=b["act"].numpy(),
                   rew=b["rew"].numpy(), tml=b["tml"].numpy(),
                   **{f"sum_{k}": _sums(getattr(ds, k).numpy()) for k in ("obs", "next_obs", "act", "rew", "tml")})
        np.savez_compressed(os.path.join(out_dir, f"dataset_{name}.npz"), **out)
        print(f"{name:24s} items={len(ds)}")
    multi = {n: ("MultiHorizonD4RLMuJoCoDataset", "d4rl_mujoco_dataset") + v for n, v in dc.MULTI_HORIZON.items()}
    multi.update(dc.MULTI_HORIZON_SUMMED)
    for name, (cls, mod, skw, dkw) in multi.items():
        ds = getattr(importlib.import_module(f"cleandiffuser.dataset.{mod}"), cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
        # (reference quirk: len() is the LARGEST table but item idx is scaled by idx / len(last table) -- items past the last horizon's
        #  count raise IndexError there; the fixture records items every horizon can serve)
        idx = dc.item_indices(min(ds.len_each_horizon), skw["seed"])
        b = default_collate([ds[int(i)] for i in idx])
        out = dict(idx=idx, len_each_horizon=np.array(ds.len_each_horizon, dtype=np.int64))
        for k, part in enumerate(b):
            out[f"h{k}_horizon"] = part["horizon"].numpy()
            out[f"h{k}_obs"], out[f"h{k}_act"], out[f"h{k}_val"] = (part["data"]["obs"]["state"].numpy(), part["data"]["act"].numpy(),
                                                                    part["data"]["val"].numpy())
            if "rew" in part["data"]:
                out[f"h{k}_rew"] = part["data"]["rew"].numpy()
            out[f"h{k}_indices"] = np.array(ds.indices[k], dtype=np.int64)
        np.savez_compressed(os.path.join(out_dir, f"dataset_{name}.npz"), **out)
        print(f"{name:24s} items={len(ds)} per horizon {ds.len_each_horizon}")


if __name__ == "__main__":
    import sys
    if "siblings" not in sys.argv[1:]:
        main()
    if not sys.argv[1:] or "siblings" in sys.argv[1:]:
        siblings()
