"""Round 5 (VERDICT r4 missing #6 / next #9): the sequence / transition datasets of the D4RL family that share D4RLMuJoCoDataset's structure --
kitchen, antmaze, maze2d, the Decision-Veteran (strided, rescaled returns) and multi-horizon classes (cleandiffuser_amd/dataset/
episode_store.py; reference dataset/d4rl_kitchen_dataset.py, d4rl_antmaze_dataset.py, d4rl_maze2d_dataset.py, d4rl_mujoco_dataset.py:232-470) -- against fixtures of
the IMPORTED reference classes (oracle/gen_golden_dataset.py:siblings).  Pure data movement: every comparison is bit-exact.  CPU: the
constructor arrays, item tables, ``__getitem__`` and the host-side loader; GPU: the batches of ``cdx_gather_windows_f32``."""
import copy
import os

import numpy as np
import pytest
import torch
from torch.utils.data import default_collate

import cleandiffuser_amd.dataset as D
from oracle import dataset_cases as dc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _fix(name):
    return dict(np.load(os.path.join(GOLDEN, f"dataset_{name}.npz")))


def _sums(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), (a * a).sum()])


def _check_sequence(name, device):
    cls, _, skw, dkw = dc.SIBLING_SCENARIOS[name]
    f = _fix(name)
    ds = getattr(D, cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
    assert len(ds) == f["indices"].shape[0] and np.array_equal(ds.indices, f["indices"])
    assert np.array_equal(ds.get_normalizer().mean, f["mean"]) and np.array_equal(ds.get_normalizer().std, f["std"])
    for k in ("seq_obs", "seq_act", "seq_rew", "seq_val"):
        assert np.array_equal(_sums(getattr(ds, k)), f[f"sum_{k}"]), (name, k)
    if "tml_and_not_timeout" in f:
        assert np.array_equal(np.asarray(ds.tml_and_not_timeout, dtype=np.int64).reshape(f["tml_and_not_timeout"].shape), f["tml_and_not_timeout"])
    T = ds.seq_obs.shape[1]
    row0 = torch.from_numpy((ds.indices[f["idx"], 0] * T + ds.indices[f["idx"], 1]).astype(np.int32)).to(device)
    b = ds.loader(32, device=device).batch_of(row0)
    assert ("tml" in b) == ("tml" in f)
    for k, v in (("obs", b["obs"]["state"]), ("act", b["act"]), ("rew", b["rew"]), ("val", b["val"])) + ((("tml", b["tml"]),) if "tml" in f else ()):
        assert v.device.type == torch.device(device).type and v.is_contiguous()
        assert v.shape == f[k].shape and np.array_equal(v.cpu().numpy(), f[k]), (name, k)
    if "paths" in f:
        assert np.array_equal(np.asarray(ds.paths, dtype=np.int64).reshape(-1, 2), f["paths"])
    return ds, f


@pytest.mark.parametrize("name", list(dc.SIBLING_SCENARIOS))
def test_sibling_sequence_dataset_matches_reference_fixture(name):
    ds, f = _check_sequence(name, "cpu")
    items = default_collate([ds[int(i)] for i in f["idx"][:8]])          # the torch Dataset protocol (DataLoader drop-in)
    for k, v in (("obs", items["obs"]["state"]), ("act", items["act"]), ("rew", items["rew"]), ("val", items["val"])) + \
            ((("tml", items["tml"]),) if "tml" in f else ()):
        assert np.array_equal(v.numpy(), f[k][:8]), (name, k)
    n = len(ds)
    ld = ds.loader(50, shuffle=True, drop_last=True, device="cpu", generator=torch.Generator().manual_seed(1))
    assert len(ld) == n // 50 and sum(b["act"].shape[0] for b in ld) == (n // 50) * 50


def _check_td(name, device):
    cls, _, skw, dkw = dc.SIBLING_TD_SCENARIOS[name]
    f = _fix(name)
    ds = getattr(D, cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
    for k in ("obs", "next_obs", "act", "rew", "tml"):
        assert np.array_equal(_sums(getattr(ds, k).numpy()), f[f"sum_{k}"]), k
    b = ds.loader(64, device=device).batch_of(torch.from_numpy(f["idx"].astype(np.int32)).to(device))
    got = {"obs": b["obs"]["state"], "next_obs": b["next_obs"]["state"], "act": b["act"], "rew": b["rew"], "tml": b["tml"]}
    for k, v in got.items():
        assert v.shape == f[k].shape and np.array_equal(v.cpu().numpy(), f[k]), (name, k)
    it = ds[int(f["idx"][3])]
    assert np.array_equal(it["next_obs"]["state"].numpy(), f["next_obs"][3]) and np.array_equal(it["rew"].numpy(), f["rew"][3])


@pytest.mark.parametrize("name", list(dc.SIBLING_TD_SCENARIOS))
def test_sibling_transition_dataset_matches_reference_fixture(name):
    _check_td(name, "cpu")
    with pytest.raises(ValueError):
        D.D4RLAntmazeTDDataset(dc.make_data(dc.SIBLING_TD_SCENARIOS["antmaze_td_cql"][2]), reward_tune="nope")


MULTI = {**{n: ("MultiHorizonD4RLMuJoCoDataset", "d4rl_mujoco_dataset") + v for n, v in dc.MULTI_HORIZON.items()}, **dc.MULTI_HORIZON_SUMMED}


def _check_multi(name, device):
    cls, _, skw, dkw = MULTI[name]
    f = _fix(name)
    ds = getattr(D, cls)(copy.deepcopy(dc.make_data(skw)), **dkw)
    assert np.array_equal(np.array(ds.len_each_horizon), f["len_each_horizon"]) and len(ds) == int(f["len_each_horizon"].max())
    for k in range(len(ds.horizons)):
        assert np.array_equal(ds.indices[k], f[f"h{k}_indices"])
    ld = ds.loader(16, device=device, shuffle=False)
    b = ld.batch_of(torch.from_numpy(f["idx"]).to(device))
    assert len(b) == len(ds.horizons)
    for k, part in enumerate(b):
        with_rew = f"h{k}_rew" in f                                       # (the kitchen / antmaze classes keep the reward window)
        assert np.array_equal(part["horizon"].cpu().numpy(), f[f"h{k}_horizon"]) and set(part["data"]) == {"obs", "act", "val"} | ({"rew"} if with_rew else set())
        for key, v in (("obs", part["data"]["obs"]["state"]), ("act", part["data"]["act"]), ("val", part["data"]["val"])) + \
                ((("rew", part["data"]["rew"]),) if with_rew else ()):
            assert v.shape == f[f"h{k}_{key}"].shape and np.array_equal(v.cpu().numpy(), f[f"h{k}_{key}"]), (k, key)
    return ds, f, ld


@pytest.mark.parametrize("name", list(MULTI))
def test_multi_horizon_dataset_matches_reference_fixture(name):
    ds, f, ld = _check_multi(name, "cpu")
    items = default_collate([ds[int(i)] for i in f["idx"][:6]])
    for k, part in enumerate(items):
        assert np.array_equal(part["data"]["obs"]["state"].numpy(), f[f"h{k}_obs"][:6]) and np.array_equal(part["horizon"].numpy(), f[f"h{k}_horizon"][:6])
        assert np.array_equal(part["data"]["val"].numpy(), f[f"h{k}_val"][:6])
    # the reference's quirk: len() counts the largest table, items past the LAST table's count cannot be served (IndexError there too)
    with pytest.raises(IndexError):
        ds[len(ds) - 1]
    assert len(ld) == min(ds.len_each_horizon) // 16 and sum(b[0]["data"]["act"].shape[0] for b in ld) == (min(ds.len_each_horizon) // 16) * 16


def test_reference_import_paths_resolve():
    from cleandiffuser_amd.dataset.d4rl_antmaze_dataset import D4RLAntmazeDataset, D4RLAntmazeTDDataset  # noqa: F401
    from cleandiffuser_amd.dataset.d4rl_kitchen_dataset import D4RLKitchenDataset, D4RLKitchenTDDataset, DV_D4RLKitchenSeqDataset  # noqa: F401
    from cleandiffuser_amd.dataset.episode_store import DV_D4RLMuJoCoSeqDataset, MultiHorizonD4RLMuJoCoDataset  # noqa: F401
    from cleandiffuser_amd.dataset.d4rl_antmaze_dataset import DV_D4RLAntmazeSeqDataset, MultiHorizonD4RLAntmazeDataset  # noqa: F401
    from cleandiffuser_amd.dataset.d4rl_kitchen_dataset import MultiHorizonD4RLKitchenDataset  # noqa: F401
    from cleandiffuser_amd.dataset.d4rl_maze2d_dataset import D4RLMaze2DTDDataset, DV_D4RLMaze2DSeqDataset  # noqa: F401


def test_dv_maze_constructors_keep_the_reference_error_behaviour():
    """reward_tune outside {"iql", "none"} -> ValueError (after the rows were built, like the reference); an antmaze episode that is
    not exactly max_path_length steps up to its timeout -> AssertionError (reference d4rl_antmaze_dataset.py:461)."""
    skw = dict(dc.SIBLING_SCENARIOS["antmaze_dv_h5_s3"][2])
    with pytest.raises(ValueError):
        D.DV_D4RLAntmazeSeqDataset(dc.make_data(skw), horizon=5, max_path_length=60, reward_tune="cql")
    with pytest.raises(AssertionError):
        D.DV_D4RLAntmazeSeqDataset(dc.make_data(skw), horizon=5, max_path_length=61)
    with pytest.raises(ValueError):
        D.DV_D4RLMaze2DSeqDataset(dc.make_data(dc.SIBLING_SCENARIOS["maze2d_dv_h6_s2"][2]), horizon=6, max_path_length=50, reward_tune="antmaze")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(dc.SIBLING_SCENARIOS))
def test_resident_sibling_sequence_batches_match_reference_fixture(name):
    ds, f = _check_sequence(name, "cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    ld = ds.loader(128, device="cuda", generator=g)
    perm = torch.randperm(len(ds), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)).cpu().numpy()
    b = next(iter(ld))
    idx = perm[:128]
    p, s = ds.indices[idx, 0], ds.indices[idx, 1]
    win = s[:, None] + ds.stride * np.arange(ds.horizon)[None]
    want = ds.seq_obs[p[:, None], win]
    if ds.learn_policy:
        want = want.copy()
        want[:, :, :2] -= want[:, :1, :2].copy()
    assert np.array_equal(b["obs"]["state"].cpu().numpy(), want) and np.array_equal(b["val"].cpu().numpy(), ds.seq_val[p, s])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(dc.SIBLING_TD_SCENARIOS))
def test_resident_sibling_transition_batches_match_reference_fixture(name):
    _check_td(name, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MULTI))
def test_resident_multi_horizon_batches_match_reference_fixture(name):
    _check_multi(name, "cuda")
