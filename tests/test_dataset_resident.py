"""SURVEY.md 8(f4), third slice: the D4RL-MuJoCo datasets with HBM-resident buffers against the reference.

Fixtures tests/golden/dataset_*.npz are outputs of the IMPORTED reference classes (oracle/gen_golden_dataset.py).  Pure data movement:
every comparison is bit-exact.  CPU tests: the plain-loop restatement against the fixtures, the product's host arrays / item table /
`__getitem__` / host-side loader against the fixtures, loader semantics.  GPU tests: the batches of `cdx_gather_windows_f32`."""
import copy
import ctypes
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, default_collate

from cleandiffuser_amd.dataset.d4rl_mujoco_dataset import D4RLMuJoCoDataset, D4RLMuJoCoTDDataset
from oracle import dataset_cases as dc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _fix(name):
    return dict(np.load(os.path.join(GOLDEN, f"dataset_{name}.npz")))


def _sums(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), (a * a).sum()])


@pytest.mark.parametrize("name", list(dc.SCENARIOS))
def test_restated_sequence_dataset_matches_reference_fixture(name):
    skw, dkw = dc.SCENARIOS[name]
    f = _fix(name)
    r = dc.restate_sequence(dc.synthetic(**skw), **dkw)
    assert np.array_equal(r["indices"], f["indices"])
    for k in ("seq_obs", "seq_act", "seq_rew", "seq_val"):
        assert np.array_equal(_sums(r[k]), f[f"sum_{k}"]), k
    items = dc.restate_items(r, f["idx"])
    for k in ("obs", "act", "rew", "val"):
        assert np.array_equal(items[k], f[k]), k


@pytest.mark.parametrize("name", list(dc.TD_SCENARIOS))
def test_restated_transition_dataset_matches_reference_fixture(name):
    skw, dkw = dc.TD_SCENARIOS[name]
    f = _fix(name)
    r = dc.restate_td(dc.synthetic(**skw), **dkw)
    for k in ("obs", "next_obs", "act", "rew", "tml"):
        assert np.array_equal(_sums(r[k]), f[f"sum_{k}"]), k
        assert np.array_equal(r[k][f["idx"]], f[k]), k


def _row0(ds, idx, device):
    return torch.from_numpy((ds.indices[idx, 0] * ds.max_path_length + ds.indices[idx, 1]).astype(np.int32)).to(device)


def _check_sequence(name, device):
    skw, dkw = dc.SCENARIOS[name]
    f = _fix(name)
    ds = D4RLMuJoCoDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
    assert len(ds) == f["indices"].shape[0] and np.array_equal(ds.indices, f["indices"])
    assert np.array_equal(ds.path_lengths, f["path_lengths"])
    assert np.array_equal(ds.get_normalizer().mean, f["mean"]) and np.array_equal(ds.get_normalizer().std, f["std"])
    for k in ("seq_obs", "seq_act", "seq_rew", "seq_val"):
        assert np.array_equal(_sums(getattr(ds, k)), f[f"sum_{k}"]), k
    ld = ds.loader(32, device=device)
    b = ld.batch_of(_row0(ds, f["idx"], device))
    assert b["obs"]["state"].device.type == torch.device(device).type
    for k, v in (("obs", b["obs"]["state"]), ("act", b["act"]), ("rew", b["rew"]), ("val", b["val"])):
        assert v.shape == f[k].shape and np.array_equal(v.cpu().numpy(), f[k]), (name, k)
    return ds, f


@pytest.mark.parametrize("name", list(dc.SCENARIOS))
def test_sequence_dataset_host_side_matches_reference_fixture(name):
    ds, f = _check_sequence(name, "cpu")
    items = default_collate([ds[int(i)] for i in f["idx"][:8]])          # the torch Dataset protocol still works (DataLoader drop-in)
    assert np.array_equal(items["obs"]["state"].numpy(), f["obs"][:8]) and np.array_equal(items["val"].numpy(), f["val"][:8])


def _check_td(name, device):
    skw, dkw = dc.TD_SCENARIOS[name]
    f = _fix(name)
    ds = D4RLMuJoCoTDDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
    for k in ("obs", "next_obs", "act", "rew", "tml"):
        assert np.array_equal(_sums(getattr(ds, k).numpy()), f[f"sum_{k}"]), k
    b = ds.loader(64, device=device).batch_of(torch.from_numpy(f["idx"].astype(np.int32)).to(device))
    got = {"obs": b["obs"]["state"], "next_obs": b["next_obs"]["state"], "act": b["act"], "rew": b["rew"], "tml": b["tml"]}
    for k, v in got.items():
        assert v.shape == f[k].shape and np.array_equal(v.cpu().numpy(), f[k]), (name, k)


@pytest.mark.parametrize("name", list(dc.TD_SCENARIOS))
def test_transition_dataset_host_side_matches_reference_fixture(name):
    _check_td(name, "cpu")


def test_loader_epochs_cover_the_dataset_like_a_dataloader():
    skw, dkw = dc.SCENARIOS["seq_h8"]
    ds = D4RLMuJoCoDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
    n = len(ds)
    ld = ds.loader(100, shuffle=True, drop_last=True, device="cpu", generator=torch.Generator().manual_seed(3))
    assert len(ld) == n // 100 == len(DataLoader(ds, batch_size=100, drop_last=True))
    # identify every yielded item by its value field + first observation: an epoch never repeats an item
    seen = torch.cat([torch.cat([b["val"], b["obs"]["state"][:, 0]], dim=1) for b in ld])
    assert seen.shape[0] == (n // 100) * 100 and torch.unique(seen, dim=0).shape[0] == seen.shape[0]
    first = [next(iter(ld))["val"] for _ in range(2)]
    assert not torch.equal(first[0], first[1])                           # a new permutation per epoch
    tail = ds.loader(100, shuffle=False, drop_last=False, device="cpu")
    batches = list(tail)
    assert len(batches) == -(-n // 100) and batches[-1]["act"].shape[0] == n - 100 * (n // 100)
    ref = default_collate([ds[i] for i in range(100)])                   # unshuffled: dataset order, as DataLoader(shuffle=False)
    assert torch.equal(batches[0]["obs"]["state"], ref["obs"]["state"]) and torch.equal(batches[0]["rew"], ref["rew"])
    with pytest.raises(ValueError):
        ds.loader(0, device="cpu")


def test_empty_datasets_behave_like_the_reference():
    """A horizon longer than the padded episodes, or no finished episode at all: zero items (as the reference), an empty epoch."""
    d = dc.synthetic(n=300, o=4, a=2, seed=7, max_len=40)
    ds = D4RLMuJoCoDataset(copy.deepcopy(d), horizon=50, max_path_length=40)
    assert len(ds) == 0 and ds.indices.shape == (0, 3) and list(ds.loader(4, device="cpu")) == []
    d["terminals"][:] = False
    d["timeouts"][:] = False
    ds = D4RLMuJoCoDataset(d, horizon=4, max_path_length=40)
    assert len(ds) == 0 and ds.seq_obs.shape == (0, 40, 4) and len(ds.loader(4, device="cpu")) == 0


def test_too_long_episode_is_rejected():
    d = dc.synthetic(n=500, o=3, a=2, seed=9, max_len=50)
    d["terminals"][:] = False
    d["timeouts"][:] = False
    d["timeouts"][-1] = True
    with pytest.raises(ValueError):
        D4RLMuJoCoDataset(d, horizon=4, max_path_length=100)


def test_gather_entry_validates_before_touching_the_device():
    from cleandiffuser_amd.engine import blocks
    lib = blocks._lib()
    a = blocks.CdxGatherArgs(batch=4, n_fields=0)
    assert lib.cdx_gather_windows_f32(ctypes.byref(a), None) == -1
    a = blocks.CdxGatherArgs(batch=0, n_fields=1)
    assert lib.cdx_gather_windows_f32(ctypes.byref(a), None) == 0          # empty batch
    a = blocks.CdxGatherArgs(batch=4, n_fields=1, row0=8, rows=10)
    a.field[0] = blocks.CdxGatherField(src=8, out=8, width=3, steps=11)      # a window longer than the buffer
    assert lib.cdx_gather_windows_f32(ctypes.byref(a), None) == -1
    assert lib.cdx_gather_windows_f32(None, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(dc.SCENARIOS))
def test_resident_sequence_batches_match_reference_fixture(name):
    ds, f = _check_sequence(name, "cuda")
    # a whole shuffled epoch through the kernel against host indexing of the same permutation
    g = torch.Generator(device="cuda").manual_seed(11)
    ld = ds.loader(256, device="cuda", generator=g)
    g2 = torch.Generator(device="cuda").manual_seed(11)
    perm = torch.randperm(len(ds), device="cuda", generator=g2).cpu().numpy()
    for i, b in enumerate(ld):
        idx = perm[i * 256:(i + 1) * 256]
        p, s = ds.indices[idx, 0], ds.indices[idx, 1]
        win = s[:, None] + np.arange(ds.horizon)[None]
        assert np.array_equal(b["obs"]["state"].cpu().numpy(), ds.seq_obs[p[:, None], win])
        assert np.array_equal(b["act"].cpu().numpy(), ds.seq_act[p[:, None], win])
        assert np.array_equal(b["rew"].cpu().numpy(), ds.seq_rew[p[:, None], win])
        assert np.array_equal(b["val"].cpu().numpy(), ds.seq_val[p, s])
        if i == 3:
            break


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(dc.TD_SCENARIOS))
def test_resident_transition_batches_match_reference_fixture(name):
    _check_td(name, "cuda")


@pytest.mark.gpu
def test_resident_loader_feeds_update():
    """The batch a pipeline builds from the loader (pipelines/diffuser_d4rl_mujoco.py:79-84: x0 = [act | obs]) goes straight into
    update(): no host tensor in between."""
    from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    skw, dkw = dc.SCENARIOS["seq_h32_hopper"]
    ds = D4RLMuJoCoDataset(copy.deepcopy(dc.synthetic(**skw)), **dkw)
    torch.manual_seed(0)
    net = JannerUNet1d(ds.o_dim + ds.a_dim, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], timestep_emb_type="positional", attention=False,
                       kernel_size=5)
    agent = DiscreteDiffusionSDE(net, None, ema_rate=0.999, device="cuda", diffusion_steps=20, predict_noise=False)
    losses = []
    for i, batch in enumerate(ds.loader(64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))):
        x = torch.cat([batch["act"].to("cuda"), batch["obs"]["state"].to("cuda")], -1)
        assert x.is_cuda and x.shape == (64, 32, ds.o_dim + ds.a_dim)
        losses.append(agent.update(x)["loss"])
        if i == 4:
            break
    assert all(np.isfinite(l) for l in losses)
