"""Dataset settings nobody wrote a fixture for: seeded random (class, synthetic data set, horizon / stride / path length / discount /
reward options) through the REAL reference classes (imported from /root/reference; skipped where the tree is absent) and through
cleandiffuser_amd/dataset: item tables, every constructor array, the normaliser, ``len`` and a handful of items must be IDENTICAL
(pure data movement: bit-exact); a setting the reference rejects must be rejected with the same exception type.  All fourteen classes of
the reference's d4rl_mujoco / d4rl_kitchen / d4rl_antmaze / d4rl_maze2d files are in the draw."""
import copy
import importlib
import random

import numpy as np
import pytest
import torch

import cleandiffuser_amd.dataset as D
from oracle import dataset_cases as dc
from oracle import ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present on this box")

SEQ = {  # class -> (reference module, data kind)
    "D4RLMuJoCoDataset": ("d4rl_mujoco_dataset", "mujoco"), "DV_D4RLMuJoCoSeqDataset": ("d4rl_mujoco_dataset", "mujoco"),
    "D4RLKitchenDataset": ("d4rl_kitchen_dataset", "mujoco"), "DV_D4RLKitchenSeqDataset": ("d4rl_kitchen_dataset", "mujoco"),
    "D4RLAntmazeDataset": ("d4rl_antmaze_dataset", "antmaze"), "DV_D4RLAntmazeSeqDataset": ("d4rl_antmaze_dataset", "antmaze_dv"),
    "DV_D4RLMaze2DSeqDataset": ("d4rl_maze2d_dataset", "maze2d"),
}
MULTI = {"MultiHorizonD4RLMuJoCoDataset": ("d4rl_mujoco_dataset", "mujoco"), "MultiHorizonD4RLKitchenDataset": ("d4rl_kitchen_dataset", "mujoco"),
         "MultiHorizonD4RLAntmazeDataset": ("d4rl_antmaze_dataset", "antmaze")}
TD = {"D4RLMuJoCoTDDataset": ("d4rl_mujoco_dataset", "mujoco"), "D4RLKitchenTDDataset": ("d4rl_kitchen_dataset", "mujoco"),
      "D4RLAntmazeTDDataset": ("d4rl_antmaze_dataset", "antmaze"), "D4RLMaze2DTDDataset": ("d4rl_maze2d_dataset", "maze2d")}


def _data(rng, kind):
    max_len = rng.choice([20, 37, 64])
    o, a = rng.randint(2, 6), rng.randint(1, 4)
    if kind == "antmaze_dv":
        return dict(n_eps=rng.randint(3, 9), o=o, a=a, seed=rng.randint(0, 999), max_len=max_len, kind=kind), max_len
    skw = dict(n=rng.randint(6 * max_len, 14 * max_len), o=o, a=a, seed=rng.randint(0, 999), max_len=max_len)
    if kind == "antmaze":
        skw["antmaze"] = True
    elif kind != "mujoco":
        skw["kind"] = kind
    return skw, max_len


def _draw(rng):
    group = rng.choice(["seq", "seq", "seq", "multi", "td"])
    cls = rng.choice(sorted({"seq": SEQ, "multi": MULTI, "td": TD}[group]))
    mod, kind = {**SEQ, **MULTI, **TD}[cls]
    skw, max_len = _data(rng, kind)
    kw = {}
    if group == "td":
        if cls == "D4RLMuJoCoTDDataset":
            kw["normalize_reward"] = rng.random() < 0.5
        elif cls != "D4RLKitchenTDDataset":
            kw["reward_tune"] = rng.choice(["iql", "cql", "antmaze", "none", "bogus"])
        return group, cls, mod, skw, kw
    # path length: mostly the data's, sometimes too short (the reference raises) or longer
    mpl = max_len if rng.random() < 0.8 else rng.choice([max_len - 3, max_len + 5])
    kw["max_path_length"] = mpl
    kw["discount"] = rng.choice([0.9, 0.99, 0.997, 1.0])
    if group == "multi":
        kw["horizons"] = tuple(sorted(rng.sample([2, 3, 5, 8, 13, 21, 40], 2)))
        if cls == "MultiHorizonD4RLMuJoCoDataset":
            kw["terminal_penalty"] = rng.choice([-100, 0, None])
        elif cls == "MultiHorizonD4RLAntmazeDataset":
            kw["noreaching_penalty"] = rng.choice([-100, -5])
        return group, cls, mod, skw, kw
    kw["horizon"] = rng.choice([1, 2, 4, 7, 16, 33, 70])
    if cls.startswith("DV_"):
        kw.update(stride=rng.choice([1, 1, 2, 3, 5]), center_mapping=rng.random() < 0.5)
    if cls == "D4RLMuJoCoDataset":
        kw["terminal_penalty"] = rng.choice([-100.0, None])
    if cls == "DV_D4RLMuJoCoSeqDataset":
        kw.update(terminal_penalty=rng.choice([-100, None]), full_traj_bonus=rng.choice([100, None]))
    if cls == "D4RLAntmazeDataset":
        kw["noreaching_penalty"] = rng.choice([-100.0, -1.0])
    if cls in ("DV_D4RLAntmazeSeqDataset", "DV_D4RLMaze2DSeqDataset"):
        kw.update(continous_reward_at_done=rng.random() < 0.5, reward_tune=rng.choice(["iql", "none", "none", "cql"]), learn_policy=rng.random() < 0.5)
        if cls == "DV_D4RLAntmazeSeqDataset":
            kw["only_learn_reached_policy"] = rng.random() < 0.3
    return group, cls, mod, skw, kw


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), what


def _flat(item):
    if isinstance(item, dict):
        return {k: _flat(v) for k, v in item.items()}
    if isinstance(item, (list, tuple)):
        return [_flat(v) for v in item]
    return item.numpy() if torch.is_tensor(item) else item


def _cmp(a, b, what):
    if isinstance(a, dict):
        assert set(a) == set(b), what
        for k in a:
            _cmp(a[k], b[k], f"{what}/{k}")
    elif isinstance(a, list):
        assert len(a) == len(b), what
        for i, (u, v) in enumerate(zip(a, b)):
            _cmp(u, v, f"{what}[{i}]")
    elif isinstance(a, np.ndarray):
        _same(a, b, what)
    else:
        assert a == b, what


@pytest.mark.parametrize("seed", range(10))
def test_random_dataset_settings_against_the_imported_reference(seed):
    ref_import.import_reference()
    rng = random.Random(4400 + seed)
    outcomes = []
    for _ in range(6):
        group, cls, mod, skw, kw = _draw(rng)
        tag = f"{cls} {skw} {kw}"
        data = dc.make_data(skw)
        ref_cls = getattr(importlib.import_module(f"cleandiffuser.dataset.{mod}"), cls)
        try:
            want = ref_cls(copy.deepcopy(data), **kw)
            n = len(want)
            probe = sorted({0, n - 1, *(rng.randrange(n) for _ in range(4))}) if n > 0 else []
            if group == "multi":                              # (the reference cannot serve items past the LAST horizon's table)
                probe = [i for i in probe if i < min(want.len_each_horizon)]
            want_items = [_flat(want[i]) for i in probe]
        except Exception as e:  # noqa: BLE001
            with pytest.raises(type(e)):
                got = getattr(D, cls)(copy.deepcopy(data), **kw)
                [got[i] for i in ([0, len(got) - 1] if len(got) else [])]
            outcomes.append("rejected " + type(e).__name__)
            continue
        got = getattr(D, cls)(copy.deepcopy(data), **kw)
        assert len(got) == n, tag
        _same(got.get_normalizer().mean, want.get_normalizer().mean, tag + ": mean")
        _same(got.get_normalizer().std, want.get_normalizer().std, tag + ": std")
        if group == "td":
            for k in ("obs", "next_obs", "act", "rew", "tml"):
                _same(getattr(got, k).numpy(), getattr(want, k).numpy(), f"{tag}: {k}")
        else:
            for k in ("seq_obs", "seq_act", "seq_rew") + (("seq_val",) if group == "seq" or cls == "MultiHorizonD4RLMuJoCoDataset" else ()):
                _same(getattr(got, k), getattr(want, k), f"{tag}: {k}")
            if group == "multi":
                for k in range(len(kw["horizons"])):
                    _same(got.indices[k], np.array(want.indices[k], dtype=np.int64).reshape(-1, 3), f"{tag}: table {k}")
            else:
                _same(got.indices, np.array(want.indices, dtype=np.int64).reshape(-1, 3), tag + ": item table")
        for i, w in zip(probe, want_items):
            _cmp(_flat(got[i]), w, f"{tag}: item {i}")
        # the resident loader's batch (host tensors here; one gather launch on the device) = the reference's collated items
        if probe:
            from torch.utils.data import default_collate
            ld = got.loader(len(probe), shuffle=False, drop_last=False, device="cpu")
            idx = torch.tensor(probe)
            if group == "multi":
                batch = ld.batch_of(idx)
            elif group == "td":
                batch = ld.batch_of(idx.int())
            else:
                T = got.seq_obs.shape[1]
                batch = ld.batch_of(torch.from_numpy((got.indices[probe, 0] * T + got.indices[probe, 1]).astype(np.int32)))
            _cmp(_flat(batch), _flat(default_collate([want[i] for i in probe])), f"{tag}: loader batch")
        outcomes.append("ok")
    assert outcomes.count("ok") >= 2, outcomes
