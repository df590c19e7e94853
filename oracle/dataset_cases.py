"""Dataset scenarios for SURVEY.md 8(f4), third slice (GPU-resident D4RL-MuJoCo buffers).  TEST INFRASTRUCTURE -- see oracle/__init__.py:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.

`synthetic(...)` draws a D4RL-shaped dictionary (no d4rl / gym here): episodes of random length ended by a terminal, a timeout, or both,
an unfinished tail, one constant observation feature (std 0 -> 1 in the normaliser).  `restate_*` are plain-loop restatements of the
reference constructors and `__getitem__` (cleandiffuser/dataset/d4rl_mujoco_dataset.py:76-151 and :197-236), pinned by
tests/golden/dataset_*.npz, which `python -m oracle.gen_golden_dataset` writes from the IMPORTED reference classes."""
import numpy as np

SCENARIOS = {
    # name: (synthetic kwargs, dataset kwargs, item indices recorded in the fixture are drawn with this seed)
    "seq_h8": (dict(n=6000, o=11, a=3, seed=0, max_len=200), dict(horizon=8, max_path_length=200, terminal_penalty=-100.0, discount=0.99)),
    "seq_h32_hopper": (dict(n=9000, o=11, a=3, seed=1, max_len=300), dict(horizon=32, max_path_length=300, terminal_penalty=-100.0, discount=0.997)),
    "seq_h1_nopenalty": (dict(n=3000, o=17, a=6, seed=2, max_len=120), dict(horizon=1, max_path_length=120, terminal_penalty=None, discount=0.9)),
    "seq_h64_short_paths": (dict(n=4000, o=5, a=2, seed=3, max_len=100), dict(horizon=64, max_path_length=100, terminal_penalty=-50.0, discount=0.99)),
}
TD_SCENARIOS = {
    "td_plain": (dict(n=5000, o=17, a=6, seed=4, max_len=150), dict(normalize_reward=False)),
    "td_normalized_reward": (dict(n=5000, o=11, a=3, seed=5, max_len=150), dict(normalize_reward=True)),
}
N_ITEMS = 96          # items recorded per fixture


def synthetic(n, o, a, seed, max_len):
    rng = np.random.default_rng(seed)
    obs = (rng.standard_normal((n, o)) * rng.uniform(0.5, 3.0, o) + rng.uniform(-2.0, 2.0, o)).astype(np.float32)
    obs[:, o // 2] = 1.25
    act = np.tanh(rng.standard_normal((n, a))).astype(np.float32)
    rew = rng.standard_normal(n).astype(np.float32)
    term, tout = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
    i = 0
    while i < n:
        length = int(rng.integers(1, max_len + 1))
        e = min(i + length - 1, n - 1)
        if e == n - 1 and rng.random() < 0.5:
            break                                       # an unfinished tail: the reference drops it
        if length == max_len:
            tout[e] = True
        elif rng.random() < 0.15:
            tout[e] = term[e] = True
        else:
            term[e] = True
        i = e + 1
    return dict(observations=obs, actions=act, rewards=rew, terminals=term, timeouts=tout,
                next_observations=np.roll(obs, -1, axis=0).copy())


def item_indices(n_items_total, seed):
    rng = np.random.default_rng(1000 + seed)
    fixed = [0, n_items_total - 1]
    return np.concatenate([fixed, rng.integers(0, n_items_total, N_ITEMS - len(fixed))]).astype(np.int64)


def _normaliser(x):                                     # GaussianNormalizer, reference utils/normalizers.py:47-61
    mean, std = np.mean(x, axis=0), np.std(x, axis=0)
    std[std == 0] = 1.0
    return mean, std


def restate_sequence(data, horizon, max_path_length, terminal_penalty, discount):
    """Step-by-step restatement of D4RLMuJoCoDataset.__init__ (reference d4rl_mujoco_dataset.py:76-133)."""
    obs = data["observations"].astype(np.float32)
    act = data["actions"].astype(np.float32)
    rew = data["rewards"].astype(np.float32)
    tout, term = data["timeouts"], data["terminals"]
    mean, std = _normaliser(obs)
    nobs = (obs - mean[None]) / std[None]
    n_paths = int(np.sum(np.logical_or(term, tout)))
    seq_obs = np.zeros((n_paths, max_path_length, obs.shape[1]), np.float32)
    seq_act = np.zeros((n_paths, max_path_length, act.shape[1]), np.float32)
    seq_rew = np.zeros((n_paths, max_path_length, 1), np.float32)
    seq_val = np.zeros((n_paths, max_path_length, 1), np.float32)
    indices, ptr, p = [], 0, 0
    for i in range(len(tout)):
        if tout[i] or term[i]:
            length = i - ptr + 1
            if term[i] and not tout[i] and terminal_penalty is not None:
                rew[i] = terminal_penalty
            seq_obs[p, :length] = nobs[ptr:i + 1]
            seq_act[p, :length] = act[ptr:i + 1]
            seq_rew[p, :length, 0] = rew[ptr:i + 1]
            for s in range(min(length - 1, max_path_length - horizon) + 1):
                indices.append((p, s, s + horizon))
            ptr, p = i + 1, p + 1
    seq_val[:, -1] = seq_rew[:, -1]
    for t in range(max_path_length - 2, -1, -1):
        seq_val[:, t] = seq_rew[:, t] + discount * seq_val[:, t + 1]
    return dict(seq_obs=seq_obs, seq_act=seq_act, seq_rew=seq_rew, seq_val=seq_val, indices=np.array(indices, np.int64).reshape(-1, 3))


def restate_items(r, idx):
    """__getitem__ + default collate (reference :138-151)."""
    out = {k: [] for k in ("obs", "act", "rew", "val")}
    for i in idx:
        p, s, e = r["indices"][i]
        out["obs"].append(r["seq_obs"][p, s:e]); out["act"].append(r["seq_act"][p, s:e])
        out["rew"].append(r["seq_rew"][p, s:e]); out["val"].append(r["seq_val"][p, s])
    return {k: np.stack(v) for k, v in out.items()}


def restate_td(data, normalize_reward):
    """D4RLMuJoCoTDDataset.__init__ (reference :197-221; reward rescaling :10-31)."""
    data = {k: v.copy() for k, v in data.items()}
    if normalize_reward:
        rets, ep_ret, ep_len = [], 0.0, 0
        for r, d in zip(data["rewards"], data["terminals"]):
            ep_ret += float(r); ep_len += 1
            if d or ep_len == 1000:
                rets.append(ep_ret); ep_ret, ep_len = 0.0, 0
        data["rewards"] /= max(rets) - min(rets)
        data["rewards"] *= 1000
    obs = data["observations"].astype(np.float32)
    mean, std = _normaliser(obs)
    return dict(obs=((obs - mean[None]) / std[None]).astype(np.float32),
                next_obs=((data["next_observations"].astype(np.float32) - mean[None]) / std[None]).astype(np.float32),
                act=data["actions"].astype(np.float32), rew=data["rewards"].astype(np.float32)[:, None],
                tml=data["terminals"].astype(np.float32)[:, None])


# ---- round 5: the siblings over the same padded-episode structure (cleandiffuser_amd/dataset/episode_store.py) -------------------------
# name: (class name in both packages, module name in the reference, synthetic kwargs, dataset kwargs)
SIBLING_SCENARIOS = {
    "kitchen_h16": ("D4RLKitchenDataset", "d4rl_kitchen_dataset", dict(n=5000, o=9, a=4, seed=6, max_len=140), dict(horizon=16, max_path_length=140, discount=0.99)),
    "kitchen_dv_h8_s3": ("DV_D4RLKitchenSeqDataset", "d4rl_kitchen_dataset", dict(n=4000, o=7, a=3, seed=7, max_len=120),
                         dict(horizon=8, max_path_length=120, discount=0.997, center_mapping=True, stride=3)),
    "antmaze_h12": ("D4RLAntmazeDataset", "d4rl_antmaze_dataset", dict(n=6000, o=8, a=3, seed=8, max_len=90, antmaze=True),
                    dict(horizon=12, max_path_length=90, noreaching_penalty=-100.0, discount=0.99)),
    "mujoco_dv_h6_s4": ("DV_D4RLMuJoCoSeqDataset", "d4rl_mujoco_dataset", dict(n=5000, o=11, a=3, seed=9, max_len=110),
                        dict(terminal_penalty=-100, horizon=6, max_path_length=110, discount=0.99, center_mapping=True, stride=4, full_traj_bonus=100)),
    "mujoco_dv_h4_s1_01": ("DV_D4RLMuJoCoSeqDataset", "d4rl_mujoco_dataset", dict(n=3000, o=5, a=2, seed=10, max_len=80),
                           dict(terminal_penalty=None, horizon=4, max_path_length=80, discount=0.9, center_mapping=False, stride=1, full_traj_bonus=None)),
}
SIBLING_TD_SCENARIOS = {
    "kitchen_td": ("D4RLKitchenTDDataset", "d4rl_kitchen_dataset", dict(n=4000, o=9, a=4, seed=11, max_len=140), dict()),
    "antmaze_td_cql": ("D4RLAntmazeTDDataset", "d4rl_antmaze_dataset", dict(n=4000, o=8, a=3, seed=12, max_len=90), dict(reward_tune="cql")),
}
MULTI_HORIZON = {"mujoco_multi_h5_20": (dict(n=6000, o=11, a=3, seed=13, max_len=150),
                                        dict(terminal_penalty=-100, horizons=(5, 20), max_path_length=150, discount=0.99))}


def synthetic_antmaze(n, o, a, seed, max_len):
    """Antmaze-shaped flags: sparse 0 / 1 rewards; an episode either fills `max_len` steps (timeout on its last step, goal never
    reached) or reaches the goal early -- then the done flag (terminal) stays up for a few steps before the next episode starts."""
    rng = np.random.default_rng(seed)
    d = synthetic(n, o, a, seed, max_len)
    term, tout, rew = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool), np.zeros(n, dtype=np.float32)
    i = 0
    while i < n:
        if rng.random() < 0.3:
            e = i + max_len - 1
            if e >= n:
                break
            tout[e] = True
            i = e + 1
        else:
            reach = i + int(rng.integers(2, max_len - 6))
            stay = int(rng.integers(1, 5))
            if reach + stay >= n:
                break
            term[reach:reach + stay] = True
            rew[reach:reach + stay] = 1.0
            i = reach + stay
    d["terminals"], d["timeouts"], d["rewards"] = term, tout, rew
    return d


def synthetic_antmaze_dv(n_eps, o, a, seed, max_len):
    """Antmaze data as Decision-Veteran's class expects it: every episode exactly `max_len` steps with a timeout on the last one; two
    thirds reach the goal at some step (reward 1 and the terminal flag from there to the end of the episode)."""
    rng = np.random.default_rng(seed)
    n = n_eps * max_len
    d = synthetic(n, o, a, seed, max_len)
    term, tout, rew = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool), np.zeros(n, dtype=np.float32)
    for e in range(n_eps):
        lo = e * max_len
        tout[lo + max_len - 1] = True
        if rng.random() < 0.67:
            reach = lo + int(rng.integers(1, max_len))
            term[reach:lo + max_len] = True
            rew[reach:lo + max_len] = 1.0
    d["terminals"], d["timeouts"], d["rewards"] = term, tout, rew
    return d


def synthetic_maze2d(n, o, a, seed, max_len):
    """Maze2d data: one long walk, reward 1.0 while the agent sits on the goal (stays of 1-6 steps), reward 0 otherwise in stretches
    of up to 1.5 x max_len steps (so some episodes get cut to their last max_len steps); the walk may start on the goal and ends in
    a stretch that never reaches it."""
    rng = np.random.default_rng(seed)
    d = synthetic(n, o, a, seed, max_len)
    rew = np.zeros(n, dtype=np.float32)
    i = int(rng.integers(0, 3))
    rew[:i] = 1.0
    while i < n:
        i += int(rng.integers(1, int(1.5 * max_len)))
        stay = int(rng.integers(1, 7))
        rew[i:i + stay] = 1.0
        i += stay
    rew[n - 7:] = 0.0
    d["rewards"] = rew
    d["terminals"], d["timeouts"] = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
    d["timeouts"][max_len - 1::max_len] = True
    return d


def make_data(skw):
    skw = dict(skw)
    kind = "antmaze" if skw.pop("antmaze", False) else skw.pop("kind", "mujoco")
    return {"antmaze": synthetic_antmaze, "antmaze_dv": synthetic_antmaze_dv, "maze2d": synthetic_maze2d, "mujoco": synthetic}[kind](**skw)


# ---- round 5, second batch: the remaining classes of the D4RL files ---------------------------------------------------------------------
SIBLING_SCENARIOS.update({
    "antmaze_dv_h5_s3": ("DV_D4RLAntmazeSeqDataset", "d4rl_antmaze_dataset", dict(n_eps=30, o=8, a=3, seed=14, max_len=60, kind="antmaze_dv"),
                         dict(horizon=5, max_path_length=60, discount=0.99, stride=3)),
    "antmaze_dv_policy": ("DV_D4RLAntmazeSeqDataset", "d4rl_antmaze_dataset", dict(n_eps=24, o=6, a=2, seed=15, max_len=50, kind="antmaze_dv"),
                          dict(horizon=4, max_path_length=50, discount=0.95, stride=2, learn_policy=True, continous_reward_at_done=True,
                               reward_tune="none", center_mapping=False)),
    "maze2d_dv_h6_s2": ("DV_D4RLMaze2DSeqDataset", "d4rl_maze2d_dataset", dict(n=3000, o=4, a=2, seed=16, max_len=50, kind="maze2d"),
                        dict(horizon=6, max_path_length=50, discount=0.99, stride=2)),
    "maze2d_dv_policy": ("DV_D4RLMaze2DSeqDataset", "d4rl_maze2d_dataset", dict(n=2030, o=4, a=2, seed=17, max_len=40, kind="maze2d"),
                         dict(horizon=3, max_path_length=40, discount=0.9, stride=4, learn_policy=True, continous_reward_at_done=True,
                              reward_tune="none", center_mapping=False)),
})
SIBLING_TD_SCENARIOS.update({
    "maze2d_td_antmaze": ("D4RLMaze2DTDDataset", "d4rl_maze2d_dataset", dict(n=3000, o=4, a=2, seed=18, max_len=50, kind="maze2d"), dict(reward_tune="antmaze")),
})
# multi-horizon classes whose items carry the reward window and sum "val" on the fly: name -> (class, module, synthetic kwargs, kwargs)
MULTI_HORIZON_SUMMED = {
    "kitchen_multi_h4_12": ("MultiHorizonD4RLKitchenDataset", "d4rl_kitchen_dataset", dict(n=4000, o=9, a=4, seed=19, max_len=120),
                            dict(horizons=(4, 12), max_path_length=120, discount=0.99)),
    "antmaze_multi_h6_15": ("MultiHorizonD4RLAntmazeDataset", "d4rl_antmaze_dataset", dict(n=5000, o=8, a=3, seed=20, max_len=80, antmaze=True),
                            dict(horizons=(6, 15), max_path_length=80, noreaching_penalty=-100, discount=0.99)),
}
