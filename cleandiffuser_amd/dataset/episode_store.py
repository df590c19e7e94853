"""Padded-episode stores behind the sequence datasets of the D4RL family, resident in HBM (third slice; round 5).

Every sequence dataset of the reference's D4RL files -- ``d4rl_mujoco_dataset.py`` (MultiHorizon / DV variants, :232-470),
``d4rl_kitchen_dataset.py``, ``d4rl_antmaze_dataset.py``, ``d4rl_maze2d_dataset.py`` -- is the same object: per episode one row of `T` steps of (normalised observation,
action, reward, discounted return), an item table (episode, first step, one-past-last step), and ``__getitem__`` = a window of that row,
possibly strided.  They differ in where an episode ends, how the row behind its last step is padded and how the return is scaled.
``EpisodeStore`` is that object: `seq_obs / seq_act / seq_rew / seq_val` (n_paths, T, .), `indices` (n_items, 3), `stride`;
the subclasses only fill the arrays (with the reference's numpy expressions step for step, so every float is the reference's -- the
fixtures tests/golden/dataset_*.npz come from the imported reference classes), and inherit

* the ``torch.utils.data.Dataset`` side (``len`` / ``__getitem__`` / ``get_normalizer``), and
* ``loader(batch_size, shuffle, drop_last, device)``: the arrays are uploaded ONCE, an epoch's permutation is drawn on the device and
  a batch is ONE ``cdx_gather_windows_f32`` launch (a strided window: the contiguous span is gathered and every stride-th row kept).
"""
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from ..utils.normalizers import GaussianNormalizer
from .base_dataset import BaseDataset
from .d4rl_mujoco_dataset import ResidentLoader, _ResidentMixin


def episode_ends(timeouts, terminals, close_tail: bool) -> np.ndarray:
    """Index of the last step of every episode: a terminal or a timeout (and, `close_tail`, the last step of the data)."""
    done = np.logical_or(np.asarray(timeouts).astype(bool), np.asarray(terminals).astype(bool))
    if close_tail and done.shape[0]:
        done = done.copy()
        done[-1] = True
    return np.flatnonzero(done)


def discounted_returns(seq_rew: np.ndarray, discount: float, steps: Optional[int] = None) -> np.ndarray:
    """val[:, t] = rew[:, t] + discount * val[:, t + 1] from the last of the first `steps` columns backwards, the float32 recursion of
    the reference (e.g. d4rl_kitchen_dataset.py:108-110); columns past `steps` keep their reward."""
    val = np.copy(seq_rew)
    steps = val.shape[1] if steps is None else steps
    for t in range(steps - 2, -1, -1):
        val[:, t] = seq_rew[:, t] + discount * val[:, t + 1]
    return val


def window_items(lengths: Sequence[int], last_start: Sequence[int], span: int) -> np.ndarray:
    """(episode, start, start + span) for start = 0 .. last_start[episode] (none when negative), episodes in order."""
    n_items = np.maximum(np.asarray(last_start, dtype=np.int64) + 1, 0)
    path = np.repeat(np.arange(len(lengths)), n_items)
    start = np.arange(int(n_items.sum())) - np.repeat(np.cumsum(n_items) - n_items, n_items)
    return np.stack([path, start, start + span], axis=1).astype(np.int64) if n_items.sum() else np.zeros((0, 3), dtype=np.int64)


class EpisodeStore(_ResidentMixin, BaseDataset):
    """See the module docstring.  Subclasses set: normalizers, o_dim, a_dim, horizon, stride (default 1), seq_obs, seq_act, seq_rew,
    seq_val, indices, path_lengths; optionally seq_tml (an item then carries "tml", the flag of its first step) and learn_policy (the
    first two observation features of a window are taken relative to its first step: the maze classes of Decision-Veteran)."""
    stride = 1
    seq_tml = None
    learn_policy = False

    def get_normalizer(self):
        return self.normalizers["state"]

    def __len__(self):
        return self.indices.shape[0]

    def __getitem__(self, idx: int):
        path, start, end = self.indices[idx]
        s = self.stride
        obs = self.seq_obs[path, start:end:s]
        if self.learn_policy:
            obs = obs.copy()
            obs[:, :2] -= obs[0, :2]
        item = {"obs": {"state": torch.tensor(obs)}, "act": torch.tensor(self.seq_act[path, start:end:s]),
                "rew": torch.tensor(self.seq_rew[path, start:end:s]), "val": torch.tensor(self.seq_val[path, start])}
        if self.seq_tml is not None:
            item["tml"] = torch.tensor(self.seq_tml[path, start])
        return item

    # ---- resident side ----
    def _host_fields(self):
        n_paths, T = self.seq_obs.shape[:2]
        rows = n_paths * T
        span = (self.horizon - 1) * self.stride + 1
        fields = {"obs": (self.seq_obs.reshape(rows, self.o_dim), span, True), "act": (self.seq_act.reshape(rows, self.a_dim), span, True),
                  "rew": (self.seq_rew.reshape(rows, 1), span, True), "val": (self.seq_val.reshape(rows, 1), 1, False)}
        if self.seq_tml is not None:
            fields["tml"] = (self.seq_tml.reshape(rows, 1), 1, False)
        row0 = self.indices[:, 0] * T + self.indices[:, 1]
        if row0.size and int((self.indices[:, 1] + span).max()) > T:
            raise ValueError("a window leaves its episode row")
        if rows >= 2 ** 31:
            raise ValueError("more than 2^31 rows: row indices are int32")
        return fields, row0, rows

    def _assemble(self, f):
        s = self.stride
        cut = (lambda t: t[:, ::s].contiguous()) if s > 1 else (lambda t: t)
        obs = cut(f["obs"])
        if self.learn_policy:
            obs[:, :, :2] -= obs[:, :1, :2].clone()                       # (the gathered batch is ours: in place)
        out = {"obs": {"state": obs}, "act": cut(f["act"]), "rew": cut(f["rew"]), "val": f["val"]}
        if "tml" in f:
            out["tml"] = f["tml"]
        return out


def _float_fields(dataset):
    return (dataset["observations"].astype(np.float32), dataset["actions"].astype(np.float32), dataset["rewards"].astype(np.float32))


class _RepeatPadded(EpisodeStore):
    """Kitchen-style rows (reference d4rl_kitchen_dataset.py:63-111): behind an episode's last step the row repeats its last
    observation and reward with zero actions; the return runs over the whole padded row."""

    def _fill(self, dataset, row_len: int, discount: float, return_steps: Optional[int] = None):
        observations, actions, rewards = _float_fields(dataset)
        self.normalizers = {"state": GaussianNormalizer(observations)}
        nobs = self.normalizers["state"].normalize(observations)
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]
        ends = episode_ends(dataset["timeouts"], dataset["terminals"], close_tail=True)
        starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.int64) if ends.size else np.zeros(0, dtype=np.int64)
        lengths = ends - starts + 1
        n = ends.shape[0]
        self.seq_obs = np.zeros((n, row_len, self.o_dim), dtype=np.float32)
        self.seq_act = np.zeros((n, row_len, self.a_dim), dtype=np.float32)
        self.seq_rew = np.zeros((n, row_len, 1), dtype=np.float32)
        for p, (s, e, ln) in enumerate(zip(starts, ends, lengths)):
            if ln > row_len:
                raise ValueError(f"an episode of {int(ln)} steps does not fit a row of {row_len}")
            self.seq_obs[p, :ln], self.seq_act[p, :ln], self.seq_rew[p, :ln, 0] = nobs[s:e + 1], actions[s:e + 1], rewards[s:e + 1]
            self.seq_obs[p, ln:], self.seq_rew[p, ln:] = nobs[e], rewards[e]
        self.seq_val = discounted_returns(self.seq_rew, discount, return_steps)
        self.path_lengths = [int(v) for v in lengths]
        early = np.logical_and(np.asarray(dataset["terminals"]).astype(bool)[ends], np.logical_not(np.asarray(dataset["timeouts"]).astype(bool)[ends]))
        self.tml_and_not_timeout = np.stack([np.flatnonzero(early), lengths[early] - 1], axis=1).astype(np.int64) if early.any() \
            else np.array([], dtype=np.int64)
        return lengths


class D4RLKitchenDataset(_RepeatPadded):
    """Sequences of `horizon` steps with obs-repeat / act-zero / reward-repeat padding (reference d4rl_kitchen_dataset.py:10-135)."""

    def __init__(self, dataset: Dict[str, np.ndarray], horizon: int = 1, max_path_length: int = 280, discount: float = 0.99):
        super().__init__()
        self.horizon = horizon
        lengths = self._fill(dataset, max_path_length, discount)
        self.indices = window_items(lengths, np.minimum(lengths - 1, max_path_length - horizon), horizon)
        self.max_path_length = max_path_length


class DV_D4RLKitchenSeqDataset(_RepeatPadded):
    """Decision-Veteran's kitchen sequences (reference d4rl_kitchen_dataset.py:322-434): rows long enough for a strided window from
    every real step, the return rescaled to [0, 1] (`center_mapping`: [-1, 1])."""

    def __init__(self, dataset: Dict[str, np.ndarray], horizon: int = 1, max_path_length: int = 280, discount: float = 0.99,
                 center_mapping: bool = True, stride: int = 1):
        super().__init__()
        self.horizon, self.stride = horizon, stride
        span = (horizon - 1) * stride + 1
        lengths = self._fill(dataset, max_path_length + span - 1, discount, return_steps=max_path_length)
        if lengths.size and lengths.max() > max_path_length:
            raise AssertionError("an episode longer than max_path_length")
        self.indices = window_items(lengths, lengths - 1, span)
        self.seq_val = (self.seq_val - self.seq_val.min()) / (self.seq_val.max() - self.seq_val.min())
        if center_mapping:
            self.seq_val = self.seq_val * 2 - 1
        self.max_path_length = max_path_length


class D4RLAntmazeDataset(EpisodeStore):
    """Antmaze sequences (reference d4rl_antmaze_dataset.py:10-139): reward - 1 per step; an episode runs up to the step BEFORE the
    one where the done flag drops again (or that follows a timeout); a short row is padded with the observation of that next step,
    zero actions and zero rewards, a full row is an episode that never reached the goal: its last reward is `noreaching_penalty`.
    The data after the last such boundary is dropped, like the reference does."""

    def __init__(self, dataset: Dict[str, np.ndarray], horizon: int = 1, max_path_length: int = 1001, noreaching_penalty: float = -100.,
                 discount: float = 0.99):
        super().__init__()
        self.horizon = horizon
        lengths = self._fill(dataset, max_path_length, noreaching_penalty)
        self.seq_val = discounted_returns(self.seq_rew, discount)
        self.indices = window_items(lengths, np.minimum(lengths - 1, max_path_length - horizon), horizon)
        self.max_path_length = max_path_length

    def _fill(self, dataset, max_path_length: int, noreaching_penalty: float):
        observations, actions, rewards = _float_fields(dataset)
        rewards -= 1
        timeouts, terminals = np.asarray(dataset["timeouts"]).astype(bool), np.asarray(dataset["terminals"]).astype(bool)
        dones = np.logical_or(timeouts, terminals)
        self.normalizers = {"state": GaussianNormalizer(observations)}
        nobs = self.normalizers["state"].normalize(observations)
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]
        cut = np.zeros(dones.shape[0], dtype=bool)                        # cut[i]: a new episode starts at step i
        cut[1:] = np.logical_or(np.logical_and(dones[:-1], np.logical_not(dones[1:])), timeouts[:-1])
        bounds = np.flatnonzero(cut)
        starts = np.concatenate([[0], bounds[:-1]]).astype(np.int64) if bounds.size else np.zeros(0, dtype=np.int64)
        lengths = bounds - starts
        n = bounds.shape[0]
        if n and lengths.max() > max_path_length:
            raise ValueError(f"path_length: {int(lengths.max())} > max_path_length: {max_path_length}")
        self.seq_obs = np.zeros((n, max_path_length, self.o_dim), dtype=np.float32)
        self.seq_act = np.zeros((n, max_path_length, self.a_dim), dtype=np.float32)
        self.seq_rew = np.zeros((n, max_path_length, 1), dtype=np.float32)
        for p, (s, b, ln) in enumerate(zip(starts, bounds, lengths)):
            self.seq_obs[p, :ln], self.seq_act[p, :ln], self.seq_rew[p, :ln, 0] = nobs[s:b], actions[s:b], rewards[s:b]
            if ln < max_path_length:
                self.seq_obs[p, ln:] = nobs[b]
            else:
                self.seq_rew[p, -1] = noreaching_penalty
        self.path_lengths = [int(v) for v in lengths]
        early = np.logical_and(terminals[bounds], np.logical_not(timeouts[bounds])) if n else np.zeros(0, dtype=bool)
        self.tml_and_not_timeout = np.stack([np.flatnonzero(early), lengths[early]], axis=1).astype(np.int64) if early.any() \
            else np.array([], dtype=np.int64)
        return lengths


class DV_D4RLMuJoCoSeqDataset(EpisodeStore):
    """Decision-Veteran's MuJoCo sequences (reference d4rl_mujoco_dataset.py:322-470): episodes end at a terminal, a timeout or the
    last step of the data; a terminal step's reward is `terminal_penalty`, the last step of an episode that fills the row gets
    `full_traj_bonus`; zero padding, one spare all-zero row; strided windows that stay inside the episode; the return rescaled to
    [0, 1] (`center_mapping`: [-1, 1])."""

    def __init__(self, dataset: Dict[str, np.ndarray], terminal_penalty: float = -100, horizon: int = 1, max_path_length: int = 1000,
                 discount: float = 0.99, center_mapping: bool = True, stride: int = 1, full_traj_bonus: float = 100):
        super().__init__()
        observations, actions, rewards = _float_fields(dataset)
        timeouts, terminals = np.asarray(dataset["timeouts"]).astype(bool), np.asarray(dataset["terminals"]).astype(bool)
        self.stride, self.horizon = stride, horizon
        self.normalizers = {"state": GaussianNormalizer(observations)}
        nobs = self.normalizers["state"].normalize(observations)
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]
        n_paths = int(np.sum(np.logical_or(terminals, timeouts)))
        ends = episode_ends(timeouts, terminals, close_tail=True)
        starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.int64) if ends.size else np.zeros(0, dtype=np.int64)
        lengths = ends - starts + 1
        if lengths.size and lengths.max() > max_path_length:
            raise AssertionError(f"current path length {int(lengths.max())}")
        if terminal_penalty is not None:
            rewards[ends[terminals[ends]]] = terminal_penalty
        if full_traj_bonus is not None:
            full = ends[lengths == max_path_length]
            rewards[full] = rewards[full] + full_traj_bonus
        self.seq_obs = np.zeros((n_paths + 1, max_path_length, self.o_dim), dtype=np.float32)
        self.seq_act = np.zeros((n_paths + 1, max_path_length, self.a_dim), dtype=np.float32)
        self.seq_rew = np.zeros((n_paths + 1, max_path_length, 1), dtype=np.float32)
        if ends.shape[0] > n_paths + 1:
            raise IndexError("more episodes than rows")                   # (cannot happen: at most one unfinished tail)
        for p, (s, e, ln) in enumerate(zip(starts, ends, lengths)):
            self.seq_obs[p, :ln], self.seq_act[p, :ln], self.seq_rew[p, :ln, 0] = nobs[s:e + 1], actions[s:e + 1], rewards[s:e + 1]
        span = (horizon - 1) * stride + 1
        self.indices = window_items(lengths, lengths - span, span)
        val = np.zeros_like(self.seq_rew)
        val[:, -1] = self.seq_rew[:, -1]
        for t in range(max_path_length - 2, -1, -1):
            val[:, t] = self.seq_rew[:, t] + discount * val[:, t + 1]
        val = (val - val.min()) / (val.max() - val.min())
        self.seq_val = val * 2 - 1 if center_mapping else val
        self.path_lengths = [int(v) for v in lengths]
        self.max_path_length = max_path_length


class MultiHorizonLoader:
    """``DataLoader`` stand-in of the multi-horizon datasets: per batch one gather launch PER HORIZON, yielding what the reference's
    collated batch is -- a list of {"horizon": (B,) int64, "data": {...}} (reference d4rl_mujoco_dataset.py:296-320)."""

    def __init__(self, ds, loaders, batch_size, shuffle, drop_last, generator, device):
        self.ds, self.loaders = ds, loaders
        self.batch_size, self.shuffle, self.drop_last, self.generator, self.device = int(batch_size), shuffle, drop_last, generator, device
        # item idx -> entry of horizon k's table: int(len_k * (idx / len_last)), the reference's float64 arithmetic -- evaluated ONCE on
        # the host for every idx (numpy float64 = Python's; a GPU's float64 division need not round the same way: found on MI355X,
        # where 1 of 96 entries came out one lower) and kept on the device as a lookup table
        lens, n = ds.len_each_horizon, self._n()
        self.sub = [torch.from_numpy((lens[k] * (np.arange(n, dtype=np.float64) / lens[-1])).astype(np.int64)).to(device)
                    for k in range(len(ds.horizons))]

    def _n(self) -> int:
        # Reference quirk (d4rl_mujoco_dataset.py:293-305): len(dataset) is the LARGEST item table, but item idx takes entry
        # int(len_k * (idx / len_last)) of table k -- past len_last the last table is indexed out of range (IndexError in the
        # reference's DataLoader).  The resident loader serves the items every horizon has.
        return min(len(self.ds), self.ds.len_each_horizon[-1])

    def __len__(self):
        n = self._n()
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def batch_of(self, idx: torch.Tensor):
        """The collated batch of dataset items `idx` (int64 on the loader's device): horizon k takes item
        int(len_k * (idx / len_last)) of its own table (the lookup table built in the constructor)."""
        out = []
        for k, (h, ld) in enumerate(zip(self.ds.horizons, self.loaders)):
            sub = self.sub[k][idx]
            out.append({"horizon": torch.full((idx.shape[0],), h, dtype=torch.int64, device=idx.device), "data": ld.batch_of(ld.item_row0[sub])})
        return out

    def __iter__(self):
        n, bs = self._n(), self.batch_size
        order = torch.randperm(n, device=self.device, generator=self.generator) if self.shuffle else torch.arange(n, device=self.device)
        for i in range(len(self)):
            yield self.batch_of(order[i * bs:min((i + 1) * bs, n)])


class _MultiHorizon(BaseDataset):
    """What the three multi-horizon datasets share (reference d4rl_mujoco_dataset.py:232-320, d4rl_kitchen_dataset.py:212-308,
    d4rl_antmaze_dataset.py:250-369): one set of episode rows, one item table per horizon, item `idx` = one window per horizon.
    Subclasses fill seq_obs / seq_act / seq_rew, call ``_tables`` and say where an item's "val" comes from."""
    with_rew = True               # do the items carry the reward window?  (the MuJoCo class drops it)

    def _tables(self, lengths, horizons, max_path_length: int):
        self.horizons = horizons
        lengths = np.asarray(lengths, dtype=np.int64)
        self.indices = [window_items(lengths, np.minimum(lengths - 1, max_path_length - h), h) for h in horizons]
        self.len_each_horizon = [int(t.shape[0]) for t in self.indices]
        self.max_path_length = max_path_length
        self._views = None

    def _item_val(self, path: int, start: int) -> np.ndarray:
        raise NotImplementedError

    def _val_rows(self) -> np.ndarray:
        """(n_paths, T, 1): the "val" of every (episode, first step) an item table can name -- what the resident loader gathers."""
        raise NotImplementedError

    def get_normalizer(self):
        return self.normalizers["state"]

    def __len__(self):
        return max(self.len_each_horizon)

    def __getitem__(self, idx: int):
        out = []
        for k, h in enumerate(self.horizons):
            path, start, end = self.indices[k][int(self.len_each_horizon[k] * (idx / self.len_each_horizon[-1]))]
            data = {"obs": {"state": torch.tensor(self.seq_obs[path, start:end])}, "act": torch.tensor(self.seq_act[path, start:end])}
            if self.with_rew:
                data["rew"] = torch.tensor(self.seq_rew[path, start:end])
            data["val"] = torch.tensor(self._item_val(path, start))
            out.append({"horizon": h, "data": data})
        return out

    def _view(self, k: int, val_rows: np.ndarray) -> EpisodeStore:
        """Horizon k as an EpisodeStore over the SAME arrays (uploaded once, by the first view)."""
        v = EpisodeStore()
        v.normalizers, v.o_dim, v.a_dim, v.horizon = self.normalizers, self.o_dim, self.a_dim, self.horizons[k]
        v.seq_obs, v.seq_act, v.seq_rew, v.seq_val, v.indices = self.seq_obs, self.seq_act, self.seq_rew, val_rows, self.indices[k]
        return v

    def loader(self, batch_size: int, shuffle: bool = True, drop_last: bool = True, device="cuda",
               generator: Optional[torch.Generator] = None) -> MultiHorizonLoader:
        if self._views is None:
            val_rows = self._val_rows()
            self._views = [self._view(k, val_rows) for k in range(len(self.horizons))]
            first = self._views[0].resident(device)                       # one upload; the other horizons share the device buffers
            for v in self._views[1:]:
                fields, row0, rows = v._host_fields()
                v._resident = {"device": first["device"], "rows": rows, "row0": torch.from_numpy(np.ascontiguousarray(row0, dtype=np.int32)).to(device),
                               "fields": {k: (first["fields"][k][0], fields[k][1], fields[k][2]) for k in fields}}
        names = ("obs", "act", "rew", "val") if self.with_rew else ("obs", "act", "val")

        def assemble(f):
            return {"obs": {"state": f["obs"]}, **{k: f[k] for k in names[1:]}}
        loaders = []
        for v in self._views:
            r = v.resident(device)
            loaders.append(ResidentLoader({k: r["fields"][k] for k in names}, r["row0"], r["rows"], batch_size, False, False, None, assemble))
        return MultiHorizonLoader(self, loaders, batch_size, shuffle, drop_last, generator, torch.device(device))


class MultiHorizonD4RLMuJoCoDataset(_MultiHorizon):
    """DiffuserLite's multi-horizon sequences (reference d4rl_mujoco_dataset.py:232-320): the episode arrays of ``D4RLMuJoCoDataset``,
    one item table per horizon; item `idx` = one window per horizon (no reward field), "val" = the episode's discounted return from
    the window's first step (the constructor's float32 recursion)."""
    with_rew = False

    def __init__(self, dataset, terminal_penalty=-100, horizons=(10, 20), max_path_length=1000, discount=0.99):
        super().__init__()
        from .d4rl_mujoco_dataset import D4RLMuJoCoDataset
        base = D4RLMuJoCoDataset(dataset, terminal_penalty=terminal_penalty, horizon=1, max_path_length=max_path_length, discount=discount)
        self.normalizers = base.normalizers
        self.o_dim, self.a_dim = base.o_dim, base.a_dim
        self.discount = discount ** np.arange(max_path_length, dtype=np.float32)
        self.seq_obs, self.seq_act, self.seq_rew, self.seq_val = base.seq_obs, base.seq_act, base.seq_rew, base.seq_val
        self.path_lengths = base.path_lengths
        self._tables(base.path_lengths, horizons, max_path_length)

    def _item_val(self, path, start):
        return self.seq_val[path, start]

    def _val_rows(self):
        return self.seq_val


class _MultiHorizonSummed(_MultiHorizon):
    """The kitchen / antmaze variants: an item's "val" is summed when the item is read -- (rew[start:] * discount[:T - start]).sum(0), a
    float32 numpy reduction (reference d4rl_kitchen_dataset.py:291-292, d4rl_antmaze_dataset.py:350-351).  ``__getitem__`` evaluates
    that expression; the resident loader evaluates it ONCE for every (episode, first step) an item table names -- the same numpy
    expression on the same rows, so the same bits -- and gathers from the table."""

    def _item_val(self, path, start):
        rew = self.seq_rew[path, start:]
        return (rew * self.discount[:rew.shape[0], None]).sum(0)

    def _val_rows(self):
        val = np.zeros_like(self.seq_rew)
        T = self.seq_rew.shape[1]
        last = np.minimum(np.asarray(self.path_lengths, dtype=np.int64) - 1, T - min(self.horizons))
        for p, hi in enumerate(last):
            for st in range(int(hi) + 1):
                val[p, st] = self._item_val(p, st)
        return val


class MultiHorizonD4RLKitchenDataset(_MultiHorizonSummed, _RepeatPadded):
    """DiffuserLite's kitchen sequences (reference d4rl_kitchen_dataset.py:212-308): the rows of ``D4RLKitchenDataset``."""

    def __init__(self, dataset, horizons=(10, 20), max_path_length=280, discount=0.99):
        BaseDataset.__init__(self)
        self.discount = discount ** np.arange(max_path_length, dtype=np.float32)
        lengths = self._fill(dataset, max_path_length, discount=0.0, return_steps=1)
        del self.seq_val                                                   # (the recursion of the single-horizon class is not this class's "val")
        self._tables(lengths, horizons, max_path_length)


class MultiHorizonD4RLAntmazeDataset(_MultiHorizonSummed, D4RLAntmazeDataset):
    """DiffuserLite's antmaze sequences (reference d4rl_antmaze_dataset.py:250-369): the rows of ``D4RLAntmazeDataset``."""

    def __init__(self, dataset, horizons=(10, 20), max_path_length=1001, noreaching_penalty=-100, discount=0.99):
        BaseDataset.__init__(self)
        self.discount = discount ** np.arange(max_path_length, dtype=np.float32)
        lengths = self._fill(dataset, max_path_length, noreaching_penalty)
        self._tables(lengths, horizons, max_path_length)


def _rescale_returns(val: np.ndarray, center_mapping: bool) -> np.ndarray:
    val = (val - val.min()) / (val.max() - val.min())
    return val * 2 - 1 if center_mapping else val


class _DVMaze(EpisodeStore):
    """Rows of the two Decision-Veteran maze datasets: `max_path_length + (horizon - 1) * stride` steps, so that a strided window fits
    behind every real step; sparse 0 / 1 rewards shifted by `reward_tune` AFTER padding (the padding's reward moves with them); return
    over the first `max_path_length` steps, rescaled to [0, 1] (`center_mapping`: [-1, 1])."""

    def _begin(self, dataset, horizon, max_path_length, stride, learn_policy):
        observations, actions, rewards = _float_fields(dataset)
        self.learn_policy, self.stride, self.horizon, self.max_path_length = learn_policy, stride, horizon, max_path_length
        self.normalizers = {"state": GaussianNormalizer(observations)}
        self.o_dim, self.a_dim = observations.shape[-1], actions.shape[-1]
        self._rows, self._row_len = [], max_path_length + (horizon - 1) * stride
        return self.normalizers["state"].normalize(observations), actions, rewards

    def _row(self, obs, act, rew, pad_obs, pad_rew: float, tml=None):
        """One episode row: the real steps, then `pad_obs` (None: zeros) / zero actions / `pad_rew` (/ raised terminal flags)."""
        L, n = self._row_len, obs.shape[0]
        o, a_, r, t = (np.zeros((L, d), dtype=np.float32) for d in (self.o_dim, self.a_dim, 1, 1))
        o[:n], a_[:n], r[:n, 0] = obs, act, rew
        if pad_obs is not None:
            o[n:], r[n:] = pad_obs, pad_rew
        if tml is not None:
            t[:n, 0] = tml
            if pad_obs is not None:
                t[n:] = 1
        self._rows.append((o, a_, r, t))
        return n

    def _finish(self, lengths, last_start, reward_tune, discount, center_mapping, with_tml=False):
        L = self._row_len
        cols = list(zip(*self._rows)) if self._rows else [[], [], [], []]
        # (np.array of an empty list of rows would lose the trailing dimensions, as in the reference; keep the shape instead)
        self.seq_obs, self.seq_act, self.seq_rew, tml = (np.array(c, dtype=np.float32).reshape(len(c), L, d)
                                                          for c, d in zip(cols, (self.o_dim, self.a_dim, 1, 1)))
        if with_tml:
            self.seq_tml = tml
        del self._rows
        if reward_tune == "iql":
            self.seq_rew += -1
        elif reward_tune != "none":
            raise ValueError(f"reward_tune: {reward_tune} is not supported.")
        self.seq_val = _rescale_returns(discounted_returns(self.seq_rew, discount, steps=self.max_path_length), center_mapping)
        self.indices = window_items(lengths, last_start, (self.horizon - 1) * self.stride + 1)
        self.path_lengths = [int(v) for v in lengths]


class DV_D4RLAntmazeSeqDataset(_DVMaze):
    """Decision-Veteran's antmaze sequences (reference d4rl_antmaze_dataset.py:371-570).  Every episode is `max_path_length` steps up
    to its timeout (asserted).  One that reaches the goal is cut at its FIRST terminal step (reward 1 there, asserted) and padded with
    that observation, zero actions, reward `continous_reward_at_done`, terminal flags 1 -- a window from each of its steps.  One that
    never does is kept only for policy learning (`learn_policy` and not `only_learn_reached_policy`): zero padding, windows that stay
    inside the episode.  Items also carry "tml", the terminal flag of the window's first step."""

    def __init__(self, dataset: Dict[str, np.ndarray], horizon: int = 1, max_path_length: int = 1001, discount: float = 0.99,
                 continous_reward_at_done: bool = False, reward_tune: str = "iql", center_mapping: bool = True, learn_policy: bool = False,
                 stride: int = 1, only_learn_reached_policy: bool = False):
        super().__init__()
        nobs, actions, rewards = self._begin(dataset, horizon, max_path_length, stride, learn_policy)
        timeouts, terminals = dataset["timeouts"].astype(np.float32), dataset["terminals"].astype(np.float32)
        lengths, last_start, ptr = [], [], 0
        for index in np.flatnonzero(timeouts == 1):
            assert index - ptr + 1 == max_path_length
            hit = np.flatnonzero(terminals[ptr:index + 1])
            if hit.size:
                end = ptr + int(hit[0])
                assert rewards[end] == 1
                n = self._row(nobs[ptr:end + 1], actions[ptr:end + 1], rewards[ptr:end + 1], nobs[end], 1 if continous_reward_at_done else 0,
                              tml=terminals[ptr:end + 1])
                lengths.append(n)
                last_start.append(n - 1)
            elif learn_policy and not only_learn_reached_policy:
                n = self._row(nobs[ptr:index + 1], actions[ptr:index + 1], rewards[ptr:index + 1], None, 0., tml=terminals[ptr:index + 1])
                lengths.append(n)
                last_start.append(max_path_length - (horizon - 1) * stride - 1)
            ptr = index + 1
        self._finish(lengths, last_start, reward_tune, discount, center_mapping, with_tml=True)


class DV_D4RLMaze2DSeqDataset(_DVMaze):
    """Decision-Veteran's maze2d sequences (reference d4rl_maze2d_dataset.py:9-204).  The data is one long walk with reward 1 while the
    agent sits on the goal; an episode = a stretch of reward-0 steps plus the first reward-1 step behind it (at most the last
    `max_path_length` of them; a trailing stretch that never reaches the goal is dropped).  `learn_policy`: fixed chunks of
    `max_path_length` steps instead.  Padding: last observation, zero actions, reward `continous_reward_at_done`; a window from every
    real step; `paths` keeps (first, last) data index per episode."""

    def __init__(self, dataset: Dict[str, np.ndarray], horizon: int = 1, max_path_length: int = 800, discount: float = 0.99,
                 continous_reward_at_done: bool = False, reward_tune: str = "iql", center_mapping: bool = True, learn_policy: bool = False,
                 stride: int = 1):
        super().__init__()
        nobs, actions, rewards = self._begin(dataset, horizon, max_path_length, stride, learn_policy)
        n = rewards.shape[0]
        if learn_policy:
            self.paths = [(s, min(s + max_path_length - 1, n - 1)) for s in range(0, n, max_path_length)]
        else:
            on_goal = rewards == 1.0
            arrive = np.flatnonzero(np.logical_and(on_goal[1:], np.logical_not(on_goal[:-1]))) + 1      # first reward-1 step of a stay
            leave = np.flatnonzero(np.logical_and(np.logical_not(on_goal[1:]), on_goal[:-1])) + 1       # first reward-0 step behind a stay
            first = np.concatenate([[0], leave]) if n and not on_goal[0] else leave                      # where the reward-0 stretches begin
            first = first[:arrive.shape[0]]                                                              # (a trailing stretch has no arrival)
            self.paths = [(max(int(s), int(e) - max_path_length + 1), int(e)) for s, e in zip(first, arrive)]
        pad = 1 if continous_reward_at_done else 0
        lengths = [self._row(nobs[s:e + 1], actions[s:e + 1], rewards[s:e + 1], nobs[e], pad) for s, e in self.paths]
        self._finish(lengths, [ln - 1 for ln in lengths], reward_tune, discount, center_mapping)


class D4RLKitchenTDDataset(_ResidentMixin, BaseDataset):
    """Kitchen transitions (reference d4rl_kitchen_dataset.py:138-216): normalised obs / next_obs, act, rew, tml."""

    def __init__(self, dataset: Dict[str, np.ndarray]):
        super().__init__()
        self._build(dataset, dataset["rewards"].astype(np.float32))

    def _build(self, dataset, rewards):
        observations = dataset["observations"].astype(np.float32)
        self.normalizers = {"state": GaussianNormalizer(observations)}
        self.obs = torch.tensor(self.normalizers["state"].normalize(observations))
        self.next_obs = torch.tensor(self.normalizers["state"].normalize(dataset["next_observations"].astype(np.float32)))
        self.act = torch.tensor(dataset["actions"].astype(np.float32))
        self.rew = torch.tensor(rewards)[:, None]
        self.tml = torch.tensor(dataset["terminals"].astype(np.float32))[:, None]
        self.size = self.obs.shape[0]
        self.o_dim, self.a_dim = observations.shape[-1], self.act.shape[-1]

    def get_normalizer(self):
        return self.normalizers["state"]

    def __len__(self):
        return self.size

    def __getitem__(self, idx: int):
        return {"obs": {"state": self.obs[idx]}, "next_obs": {"state": self.next_obs[idx]}, "act": self.act[idx], "rew": self.rew[idx],
                "tml": self.tml[idx]}

    def _host_fields(self):
        if self.size >= 2 ** 31:
            raise ValueError("more than 2^31 rows: row indices are int32")
        return {k: (getattr(self, k).numpy(), 1, False) for k in ("obs", "next_obs", "act", "rew", "tml")}, np.arange(self.size), self.size

    @staticmethod
    def _assemble(f):
        return {"obs": {"state": f["obs"]}, "next_obs": {"state": f["next_obs"]}, "act": f["act"], "rew": f["rew"], "tml": f["tml"]}


class D4RLAntmazeTDDataset(D4RLKitchenTDDataset):
    """Antmaze transitions with the reward tuning of the offline-RL baselines (reference d4rl_antmaze_dataset.py:142-233)."""

    def __init__(self, dataset: Dict[str, np.ndarray], reward_tune: str = "iql"):
        BaseDataset.__init__(self)
        rewards = dataset["rewards"].astype(np.float32)
        if reward_tune == "iql":
            rewards = rewards - 1.
        elif reward_tune == "cql":
            rewards = (rewards - 0.5) * 4.
        elif reward_tune == "antmaze":
            rewards = (rewards - 0.25) * 2.
        elif reward_tune != "none":
            raise ValueError(f"reward_tune: {reward_tune} is not supported.")
        self._build(dataset, rewards)


class D4RLMaze2DTDDataset(D4RLAntmazeTDDataset):
    """Maze2d transitions (reference d4rl_maze2d_dataset.py:206-290): the antmaze class's reward tunings on maze2d data."""
