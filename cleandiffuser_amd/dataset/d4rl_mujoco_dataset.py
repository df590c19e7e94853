"""D4RL-MuJoCo datasets whose buffers live in HBM (third slice).

Reference: cleandiffuser/dataset/d4rl_mujoco_dataset.py -- ``D4RLMuJoCoDataset`` (:34-151, sequences of `horizon` steps out of padded
episode arrays, Monte Carlo returns) and ``D4RLMuJoCoTDDataset`` (:154-236, transitions).  The reference pipelines wrap them in a torch
``DataLoader(batch_size, shuffle=True, num_workers=4, pin_memory=True, drop_last=True)`` and copy every batch to the device
(pipelines/diffuser_d4rl_mujoco.py:34-35,79-83, pipelines/dql_d4rl_mujoco.py:34-35,72-77): per training step ~batch_size Python
``__getitem__`` calls, a collate, a pinned copy and an H2D transfer in front of a ~10 ms ``update()``.

Here the constructor takes the same dictionary and builds the same arrays (same numpy expressions, so the floats are the reference's
bit for bit -- tests/golden/dataset_*.npz come from the imported reference), and the classes stay valid ``torch.utils.data.Dataset``s
(``len`` / ``__getitem__`` / ``get_normalizer`` as in the reference).  What is new is ``loader(...)``: the arrays are uploaded ONCE,
an epoch's permutation is drawn on the device, and every batch is ONE launch of ``cdx_gather_windows_f32`` (csrc/cdx_train.hip) that
copies each item's window -- `horizon` consecutive rows, one contiguous segment -- for all fields at once.  The loader yields the
dictionary a collated reference batch has, already on the device, so a pipeline's ``batch["act"].to(device)`` is a no-op.

The siblings over the same structure -- multi-horizon / Decision-Veteran MuJoCo sequences, the D4RL kitchen and antmaze classes -- live in
``episode_store.py`` (round 5).  The datasets that need simulators or zarr stores to exist at all (robomimic, push-T, the relay-kitchen
demos) stay out of scope.
"""
from typing import Dict, Optional

import numpy as np
import torch

from ..utils.normalizers import GaussianNormalizer
from .base_dataset import BaseDataset

__all__ = ["D4RLMuJoCoDataset", "D4RLMuJoCoTDDataset", "ResidentLoader", "return_reward_range", "modify_reward"]


def return_reward_range(dataset, max_episode_steps):
    """(min, max) episodic return; an episode ends at a terminal or after `max_episode_steps` steps, the unfinished tail does not
    count (reference d4rl_mujoco_dataset.py:10-23)."""
    rewards = np.asarray(dataset["rewards"], dtype=np.float64)
    terminals = np.asarray(dataset["terminals"]).astype(bool)
    returns, ep_ret, ep_len = [], 0.0, 0
    for r, d in zip(rewards.tolist(), terminals.tolist()):          # (the running float64 sum of the reference, step by step)
        ep_ret += r
        ep_len += 1
        if d or ep_len == max_episode_steps:
            returns.append(ep_ret)
            ep_ret, ep_len = 0.0, 0
    return min(returns), max(returns)


def modify_reward(dataset, max_episode_steps=1000):
    """Rescale the rewards IN PLACE so that episodic returns span `max_episode_steps` (reference :26-31)."""
    min_ret, max_ret = return_reward_range(dataset, max_episode_steps)
    dataset["rewards"] /= max_ret - min_ret
    dataset["rewards"] *= max_episode_steps
    return dataset


class ResidentLoader:
    """What ``DataLoader(dataset, batch_size, shuffle, drop_last)`` is to a pipeline -- an iterable of collated batches, one epoch per
    ``iter()`` -- over buffers that already sit on `device`.  `fields`: name -> (2-D fp32 matrix of `rows` rows, steps, keep_steps_dim);
    `item_row0`: int32 vector, first row of every dataset item; a batch = one gather launch over all fields.

    Shuffling uses ``torch.randperm`` on the device (pass a `generator` of that device for reproducible epochs); the ORDER of a
    shuffled epoch therefore differs from the reference DataLoader's host permutation -- the items of a batch, given its indices, do
    not (``batch_of``)."""

    def __init__(self, fields, item_row0: torch.Tensor, rows: int, batch_size: int, shuffle: bool, drop_last: bool,
                 generator: Optional[torch.Generator], assemble):
        self.fields, self.item_row0, self.rows = fields, item_row0, rows
        self.batch_size, self.shuffle, self.drop_last, self.generator = int(batch_size), shuffle, drop_last, generator
        self._assemble = assemble
        if self.batch_size <= 0:
            raise ValueError("batch_size must be positive")

    def __len__(self):
        n = self.item_row0.shape[0]
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def batch_of(self, row0: torch.Tensor):
        """The collated batch of the items whose first rows are `row0` (int32, on the loader's device)."""
        from ..engine import blocks
        names = list(self.fields)
        specs = [(self.fields[k][0], self.fields[k][1]) for k in names]
        if row0.is_cuda:
            outs = blocks.gather_windows(row0, specs, self.rows)               # raises if libcdx.so is missing: no silent fallback
        else:                                                                  # host tensors (the reference's own device): plain indexing
            outs = [src[(row0.long()[:, None] + torch.arange(steps)[None, :])] for src, steps in specs]
        flat = {k: (o if self.fields[k][2] else o[:, 0]) for k, o in zip(names, outs)}
        return self._assemble(flat)

    def __iter__(self):
        n, bs = self.item_row0.shape[0], self.batch_size
        dev = self.item_row0.device
        if self.shuffle:
            perm = torch.randperm(n, device=dev, generator=self.generator)
            epoch = self.item_row0[perm]                                       # one index launch per EPOCH
        else:
            epoch = self.item_row0
        for i in range(len(self)):
            yield self.batch_of(epoch[i * bs:min((i + 1) * bs, n)])


class _ResidentMixin:
    """Upload-once bookkeeping shared by the two datasets."""
    _resident = None

    def _host_fields(self):
        raise NotImplementedError

    def resident(self, device) -> dict:
        """The buffers as fp32 matrices on `device` (uploaded on first use, then kept)."""
        device = torch.device(device)
        if self._resident is None or self._resident["device"] != device:
            fields, row0, rows = self._host_fields()
            self._resident = {
                "device": device, "rows": rows,
                "fields": {k: (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device), steps, keep)
                           for k, (a, steps, keep) in fields.items()},
                "row0": torch.from_numpy(np.ascontiguousarray(row0, dtype=np.int32)).to(device)}
        return self._resident

    def resident_bytes(self) -> int:
        fields, row0, _ = self._host_fields()
        return int(sum(a.size * 4 for a, _, _ in fields.values()) + row0.size * 4)

    def loader(self, batch_size: int, shuffle: bool = True, drop_last: bool = True, device="cuda",
               generator: Optional[torch.Generator] = None) -> ResidentLoader:
        r = self.resident(device)
        return ResidentLoader(r["fields"], r["row0"], r["rows"], batch_size, shuffle, drop_last, generator, self._assemble)


class D4RLMuJoCoDataset(_ResidentMixin, BaseDataset):
    """Sequences of `horizon` steps, no padding across episode ends (reference d4rl_mujoco_dataset.py:34-151).

    batch["obs"]["state"] (B, horizon, o_dim) normalised observations, batch["act"] (B, horizon, a_dim), batch["rew"] (B, horizon, 1),
    batch["val"] (B, 1) discounted return from the window's first step to the end of the padded episode.

    The constructor's arrays (`seq_obs`, `seq_act`, `seq_rew`, `seq_val`, `indices`, `path_lengths`, `tml_and_not_timeout`) are the
    reference's; they are filled with index arithmetic over all steps at once instead of a Python loop over the steps."""

    def __init__(self, dataset: Dict[str, np.ndarray], terminal_penalty: float = -100., horizon: int = 1,
                 max_path_length: int = 1000, discount: float = 0.99):
        super().__init__()
        observations = dataset["observations"].astype(np.float32)
        actions = dataset["actions"].astype(np.float32)
        rewards = dataset["rewards"].astype(np.float32)
        timeouts = np.asarray(dataset["timeouts"]).astype(bool)
        terminals = np.asarray(dataset["terminals"]).astype(bool)
        self.normalizers = {"state": GaussianNormalizer(observations)}
        normed_observations = self.normalizers["state"].normalize(observations)

        self.horizon = horizon
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]

        ends = np.flatnonzero(np.logical_or(terminals, timeouts))              # last step of every finished episode
        n_paths = ends.shape[0]
        starts = np.concatenate([[0], ends[:-1] + 1]) if n_paths else np.zeros(0, dtype=np.int64)
        lengths = ends - starts + 1
        if n_paths and lengths.max() > max_path_length:
            raise ValueError(f"an episode of {int(lengths.max())} steps does not fit max_path_length = {max_path_length}")
        early = np.logical_and(terminals[ends], np.logical_not(timeouts[ends]))  # ended by a terminal state, not by the time limit
        if terminal_penalty is not None:
            rewards[ends[early]] = terminal_penalty
        self.tml_and_not_timeout = np.stack([np.flatnonzero(early), lengths[early] - 1], axis=1).astype(np.int64) \
            if early.any() else np.array([], dtype=np.int64)

        self.seq_obs = np.zeros((n_paths, max_path_length, self.o_dim), dtype=np.float32)
        self.seq_act = np.zeros((n_paths, max_path_length, self.a_dim), dtype=np.float32)
        self.seq_rew = np.zeros((n_paths, max_path_length, 1), dtype=np.float32)
        self.seq_val = np.zeros((n_paths, max_path_length, 1), dtype=np.float32)
        used = int(ends[-1]) + 1 if n_paths else 0                             # steps behind the last episode end are dropped
        path_of = np.repeat(np.arange(n_paths), lengths)
        pos_of = np.arange(used) - np.repeat(starts, lengths)
        self.seq_obs[path_of, pos_of] = normed_observations[:used]
        self.seq_act[path_of, pos_of] = actions[:used]
        self.seq_rew[path_of, pos_of, 0] = rewards[:used]

        # items: (path, start, start + horizon) for start = 0 .. min(length - 1, max_path_length - horizon)
        n_items = np.maximum(np.minimum(lengths - 1, max_path_length - horizon) + 1, 0)
        item_path = np.repeat(np.arange(n_paths), n_items)
        item_start = np.arange(int(n_items.sum())) - np.repeat(np.cumsum(n_items) - n_items, n_items)
        self.indices = np.stack([item_path, item_start, item_start + horizon], axis=1).astype(np.int64) \
            if n_items.sum() else np.zeros((0, 3), dtype=np.int64)

        if n_paths:
            self.seq_val[:, -1] = self.seq_rew[:, -1]
            for i in range(max_path_length - 1):
                self.seq_val[:, -2 - i] = self.seq_rew[:, -2 - i] + discount * self.seq_val[:, -1 - i]
        self.path_lengths = lengths.astype(np.int64)
        self.max_path_length = max_path_length

    def get_normalizer(self):
        return self.normalizers["state"]

    def __len__(self):
        return self.indices.shape[0]

    def __getitem__(self, idx: int):
        path_idx, start, end = self.indices[idx]
        return {"obs": {"state": torch.tensor(self.seq_obs[path_idx, start:end])},
                "act": torch.tensor(self.seq_act[path_idx, start:end]),
                "rew": torch.tensor(self.seq_rew[path_idx, start:end]),
                "val": torch.tensor(self.seq_val[path_idx, start])}

    # ---- resident side ----
    def _host_fields(self):
        n_paths, T = self.seq_obs.shape[:2]
        rows = n_paths * T
        fields = {"obs": (self.seq_obs.reshape(rows, self.o_dim), self.horizon, True),
                  "act": (self.seq_act.reshape(rows, self.a_dim), self.horizon, True),
                  "rew": (self.seq_rew.reshape(rows, 1), self.horizon, True),
                  "val": (self.seq_val.reshape(rows, 1), 1, False)}
        row0 = self.indices[:, 0] * T + self.indices[:, 1]
        if row0.size and int(row0.max()) + self.horizon > rows:
            raise ValueError("a window leaves the episode arrays")             # (cannot happen: start <= max_path_length - horizon)
        if rows >= 2 ** 31:
            raise ValueError("more than 2^31 rows: row indices are int32")
        return fields, row0, rows

    @staticmethod
    def _assemble(f):
        return {"obs": {"state": f["obs"]}, "act": f["act"], "rew": f["rew"], "val": f["val"]}


class D4RLMuJoCoTDDataset(_ResidentMixin, BaseDataset):
    """Transitions (reference d4rl_mujoco_dataset.py:154-236): batch["obs"]["state"], batch["next_obs"]["state"] (B, o_dim) normalised,
    batch["act"] (B, a_dim), batch["rew"], batch["tml"] (B, 1)."""

    def __init__(self, dataset: Dict[str, np.ndarray], normalize_reward: bool = False):
        super().__init__()
        if normalize_reward:
            dataset = modify_reward(dataset, 1000)
        observations = dataset["observations"].astype(np.float32)
        actions = dataset["actions"].astype(np.float32)
        next_observations = dataset["next_observations"].astype(np.float32)
        rewards = dataset["rewards"].astype(np.float32)
        terminals = dataset["terminals"].astype(np.float32)
        self.normalizers = {"state": GaussianNormalizer(observations)}
        self.obs = torch.tensor(self.normalizers["state"].normalize(observations), dtype=torch.float32)
        self.next_obs = torch.tensor(self.normalizers["state"].normalize(next_observations), dtype=torch.float32)
        self.act = torch.tensor(actions, dtype=torch.float32)
        self.rew = torch.tensor(rewards, dtype=torch.float32)[:, None]
        self.tml = torch.tensor(terminals, dtype=torch.float32)[:, None]
        self.size = self.obs.shape[0]
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]

    def get_normalizer(self):
        return self.normalizers["state"]

    def __len__(self):
        return self.size

    def __getitem__(self, idx: int):
        return {"obs": {"state": self.obs[idx]}, "next_obs": {"state": self.next_obs[idx]},
                "act": self.act[idx], "rew": self.rew[idx], "tml": self.tml[idx]}

    def _host_fields(self):
        if self.size >= 2 ** 31:
            raise ValueError("more than 2^31 rows: row indices are int32")
        fields = {k: (getattr(self, k).numpy(), 1, False) for k in ("obs", "next_obs", "act", "rew", "tml")}
        return fields, np.arange(self.size), self.size

    @staticmethod
    def _assemble(f):
        return {"obs": {"state": f["obs"]}, "next_obs": {"state": f["next_obs"]}, "act": f["act"], "rew": f["rew"], "tml": f["tml"]}


def __getattr__(name):
    """The multi-horizon / Decision-Veteran MuJoCo classes of the reference's d4rl_mujoco_dataset.py (:232-470) live in episode_store.py
    (which builds on this module): resolved on first use, so that ``from ...d4rl_mujoco_dataset import MultiHorizonD4RLMuJoCoDataset``
    keeps working."""
    if name in ("DV_D4RLMuJoCoSeqDataset", "MultiHorizonD4RLMuJoCoDataset"):
        from . import episode_store
        return getattr(episode_store, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
