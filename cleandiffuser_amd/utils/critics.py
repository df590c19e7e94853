"""Value heads and generic transformer pieces the pipelines evaluate on freshly sampled tensors: candidate
re-weighting critics (``DQLCritic``, IQL's ``TwinQ`` / ``V``), Diffusion Veteran's horizon critic, and the small transformer
toolkit.  Interface / checkpoint contract: reference utils/building_blocks.py:79-380 and utils/iql.py:7-95 (same attribute
names, hence the same ``state_dict`` keys).

Execution: every head here is a chain of ``Linear -> [LayerNorm] -> activation`` over rows, so on a ROCm device without
autograd the chain runs through ``engine/heads.py`` (fp32-MFMA GEMM with fused bias/activation epilogue + one fused
LayerNorm/activation launch per layer); with autograd on, or on CPU, the stock modules run.
"""
from copy import deepcopy

import einops
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .embeddings import SinusoidalEmbedding


def _rows(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    from ..engine import heads
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
        self.optimV = torch.optim.Adam(self.V.parameters(), lr=3e-4)
        self.optimQ = torch.optim.Adam(self.Q.parameters(), lr=3e-4)

    def update_target(self, mu=0.995):
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


class DVTransformerBlock(nn.Module):
    def __init__(self, hidden_size: int, n_heads: int, dropout: float = 0.0, norm_type="post"):
        super().__init__()
        self.norm_type = norm_type
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = nn.MultiheadAttention(hidden_size, n_heads, dropout, batch_first=True)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(hidden_size, hidden_size * 4), nn.GELU(approximate="tanh"),
                                 nn.Dropout(dropout), nn.Linear(hidden_size * 4, hidden_size))

    def forward(self, x: torch.Tensor):
        if self.norm_type == "post":
            x = self.norm1(x + self.attn(x, x, x)[0])
            return self.norm2(x + self.mlp(x))
        if self.norm_type == "pre":
            x = self.norm1(x)
            x = x + self.attn(x, x, x)[0]
            return x + self.mlp(self.norm2(x))
        raise NotImplementedError


class DVHorizonCritic(nn.Module):
    """Transformer value function over a planned trajectory; the value is read from token 0."""

    def __init__(self, in_dim: int, emb_dim: int, d_model: int = 384, n_heads: int = 6, depth: int = 12,
                 dropout: float = 0.0, norm_type: str = "post"):
        super().__init__()
        self.in_dim, self.emb_dim, self.d_model = in_dim, emb_dim, d_model
        self.x_proj = nn.Linear(in_dim, d_model)
        self.pos_emb = SinusoidalEmbedding(d_model)
        self.pos_emb_cache = None
        self.blocks = nn.ModuleList([DVTransformerBlock(d_model, n_heads, dropout, norm_type) for _ in range(depth)])
        self.final_layer = nn.Linear(d_model, 1)
        self.initialize_weights()

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x: torch.Tensor):
        """x (b, horizon, in_dim) -> (b, 1)."""
        if self.pos_emb_cache is None or self.pos_emb_cache.shape[0] != x.shape[1]:
            self.pos_emb_cache = self.pos_emb(torch.arange(x.shape[1], device=x.device))
        h = self.x_proj(x) + self.pos_emb_cache[None, ]
        for block in self.blocks:
            h = block(h)
        return self.final_layer(h)[:, 0, :]


class PreNorm(nn.Module):
    """fn(LayerNorm(x), **kwargs)"""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(self.norm(x), **kwargs)


class Residual(nn.Module):
    """fn(x, **kwargs) + x"""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        return self.fn(x, **kwargs) + x


class FeedForward(nn.Module):
    def __init__(self, d_model: int, hidden_scale: int = 4, dropout: float = 0.0):
        super().__init__()
        hidden = int(d_model * hidden_scale)
        self.net = nn.Sequential(nn.Linear(d_model, hidden), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden, d_model),
                                 nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class MultiHeadAttention(nn.Module):
    """Attention WITHOUT an output projection; returns (context, detached attention map laid out (b, i, j, h))."""

    def __init__(self, d_model: int, nhead: int, dropout: float = 0.1, bias: bool = False):
        super().__init__()
        assert d_model % nhead == 0, "`d_model` must be divisible by `nhead`."
        self.nhead, self.d_k = nhead, d_model // nhead
        self.scale = 1 / np.sqrt(self.d_k)
        self.q_layer = nn.Linear(d_model, d_model, bias=bias)
        self.k_layer = nn.Linear(d_model, d_model, bias=bias)
        self.v_layer = nn.Linear(d_model, d_model, bias=True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, q, k, v, mask=None):
        if mask is not None:
            if mask.dim() == 2:
                assert mask.shape == (q.shape[1], k.shape[1])
                mask = mask.unsqueeze(0)
            elif mask.dim() == 3:
                assert mask.shape == (q.shape[0], q.shape[1], k.shape[1])
            else:
                raise ValueError("`mask` shape should be either (i, j) or (b, i, j)")
            mask = mask.unsqueeze(-1)
        split = "b n (h d) -> b n h d"
        q = einops.rearrange(self.q_layer(q), split, h=self.nhead)
        k = einops.rearrange(self.k_layer(k), split, h=self.nhead)
        v = einops.rearrange(self.v_layer(v), split, h=self.nhead)
        scores = torch.einsum("b i h d, b j h d -> b i j h", q, k) * self.scale
        if mask is not None:
            scores.masked_fill_(mask == 0, float("-inf"))
        attn = self.dropout(torch.softmax(scores, dim=2))
        out = torch.einsum("b i j h, b j h d -> b i h d", attn, v)
        return einops.rearrange(out, "b i h d -> b i (h d)"), attn.detach()


def generate_causal_mask(length: int, device: torch.device = "cpu"):
    return torch.tril(torch.ones(length, length, device=device), diagonal=0)


class Transformer(nn.Module):
    """Pre-norm encoder stack; ``layers.{i}`` = [LayerNorm, MultiHeadAttention, LayerNorm, FeedForward]."""

    def __init__(self, d_model: int, nhead: int, num_layers: int, hidden_scale: int = 4, attn_dropout: float = 0.0,
                 ffn_dropout: float = 0.0, bias: bool = False):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([nn.LayerNorm(d_model), MultiHeadAttention(d_model, nhead, attn_dropout, bias),
                           nn.LayerNorm(d_model), FeedForward(d_model, hidden_scale, ffn_dropout)])
            for _ in range(num_layers)])

    def forward(self, x, mask=None):
        maps = []
        for norm1, attn, norm2, ffn in self.layers:
            h = norm1(x)
            h, amap = attn(h, h, h, mask=mask)
            maps.append(amap)
            x = h + x
            x = ffn(norm2(x)) + x
        return x, maps
