"""Value heads the pipelines evaluate on freshly sampled tensors: the candidate re-weighting critics ``DQLCritic`` and
IQL's ``TwinQ`` / ``V``.  Interface / checkpoint contract: reference utils/building_blocks.py:79-147 and utils/iql.py:7-95 (same
attribute names, hence the same ``state_dict`` keys).  Diffusion Veteran's horizon critic and the reference's transformer toolkit
(building_blocks.py:149-373) are outside the sampling path and are not mirrored.

Execution: every head here is a chain of ``Linear -> [LayerNorm] -> activation`` over rows, so on a ROCm device without
autograd the chain runs through ``engine/heads.py`` (fp32-MFMA GEMM with fused bias/activation epilogue + one fused
LayerNorm/activation launch per layer); with autograd ON (the critics' training steps) through ``engine/train.py:chain_forward`` --
the same library nodes the denoisers train on, forward and backward --, IQL's two Adam optimisers and its Polyak target update on
``cdx_optim_f32``; on CPU the stock modules run.
"""
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F


def _rows(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    from ..engine import heads, train
    if train.supports_chain(seq, x):           # autograd on, ROCm device (the critics' own training steps): library nodes forward and backward
        return train.chain_forward(seq, x)
    y = heads.try_sequential(seq, x)
    return seq(x) if y is None else y


class SoftLowerBound(nn.Module):
    """x -> lb + softplus(x - lb): smooth clamp from below."""

    def __init__(self, lower_bound: float):
        super().__init__()
        self.lower_bound = lower_bound

    def forward(self, x):
        return self.lower_bound + F.softplus(x - self.lower_bound)


class SoftUpperBound(nn.Module):
    """x -> ub - softplus(ub - x): smooth clamp from above."""

    def __init__(self, upper_bound: float):
        super().__init__()
        self.upper_bound = upper_bound

    def forward(self, x):
        return self.upper_bound - F.softplus(self.upper_bound - x)


def _q_tower(in_dim: int, hidden: int, first_act: nn.Module, depth: int) -> nn.Sequential:
    layers, acts = [], [first_act] + [nn.Mish() for _ in range(depth - 1)]
    for i, act in enumerate(acts):
        layers += [nn.Linear(in_dim if i == 0 else hidden, hidden), nn.LayerNorm(hidden), act]
    return nn.Sequential(*layers, nn.Linear(hidden, 1))


class DQLCritic(nn.Module):
    """Double-Q critic of Diffusion-QL: two towers Linear-LN-Tanh, 2x(Linear-LN-Mish), Linear(1)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden_dim: int = 256):
        super().__init__()
        self.q1_model = _q_tower(obs_dim + act_dim, hidden_dim, nn.Tanh(), 3)
        self.q2_model = _q_tower(obs_dim + act_dim, hidden_dim, nn.Tanh(), 3)

    def forward(self, obs, act):
        x = torch.cat([obs, act], dim=-1)
        return _rows(self.q1_model, x), _rows(self.q2_model, x)

    def q1(self, obs, act):
        return _rows(self.q1_model, torch.cat([obs, act], dim=-1))

    def q_min(self, obs, act):
        return torch.min(*self.forward(obs, act))


class TwinQ(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden_dim: int = 256):
        super().__init__()
        self.Q1 = _q_tower(obs_dim + act_dim, hidden_dim, nn.Mish(), 2)
        self.Q2 = _q_tower(obs_dim + act_dim, hidden_dim, nn.Mish(), 2)

    def both(self, obs, act):
        x = torch.cat([obs, act], -1)
        return _rows(self.Q1, x), _rows(self.Q2, x)

    def forward(self, obs, act):
        return torch.min(*self.both(obs, act))


class V(nn.Module):
    def __init__(self, obs_dim, hidden_dim: int = 256):
        super().__init__()
        self.V = _q_tower(obs_dim, hidden_dim, nn.Mish(), 2)

    def forward(self, obs):
        return _rows(self.V, obs)


IDQLQNet = TwinQ
IDQLVNet = V


class IQL(nn.Module):
    """Implicit Q-Learning: expectile-regressed V, TD-regressed twin Q with a Polyak target (reference utils/iql.py:40-95)."""

    def __init__(self, obs_dim: int, act_dim: int, tau: float = 0.7, discount: float = 0.99, hidden_dim: int = 256):
        super().__init__()
        self.iql_tau, self.discount = tau, discount
        self.Q = TwinQ(obs_dim, act_dim, hidden_dim)
        self.Q_targ = deepcopy(self.Q).requires_grad_(False).eval()
        self.V = V(obs_dim, hidden_dim)
        # (torch.optim.Adam subclasses: one multi-tensor launch of cdx_optim_f32 per step on a ROCm device, torch's own step elsewhere)
        from ..engine.optim import FusedAdam
        self.optimV = FusedAdam(self.V.parameters(), lr=3e-4)
        self.optimQ = FusedAdam(self.Q.parameters(), lr=3e-4)

    def update_target(self, mu=0.995):
        from ..engine.optim import ema_update_native
        if ema_update_native(self.Q, self.Q_targ, mu):       # targ <- mu * targ + (1 - mu) * q over every pair, one launch
            return
        for p, p_targ in zip(self.Q.parameters(), self.Q_targ.parameters()):
            p_targ.data = mu * p_targ.data + (1 - mu) * p.data

    def update_V(self, obs, act):
        adv = self.Q_targ(obs, act) - self.V(obs)
        loss = (torch.abs(self.iql_tau - (adv < 0).float()) * adv ** 2).mean()
        self.optimV.zero_grad()
        loss.backward()
        self.optimV.step()
        return loss.item()

    def update_Q(self, obs, act, rew, obs_next, done):
        with torch.no_grad():
            target = rew + self.discount * (1 - done) * self.V(obs_next)
        q1, q2 = self.Q.both(obs, act)
        loss = ((q1 - target) ** 2 + (q2 - target) ** 2).mean()
        self.optimQ.zero_grad()
        loss.backward()
        self.optimQ.step()
        self.update_target()
        return loss.item()

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path, device):
        self.load_state_dict(torch.load(path, map_location=device))
