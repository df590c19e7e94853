"""MLP denoiser backbones for low-dimensional x of shape (b, x_dim): PearceMlp (DiffusionBC), DQLMlp (Diffusion-QL),
IDQLMlp / NewIDQLMlp (IDQL, SynthER's "ResidualMLP"), MlpNNDiffusion.

Interface + checkpoint contract = reference nn_diffusion/{pearcemlp.py:36-79, dqlmlp.py:9-52, idqlmlp.py:21-112,
mlps.py:10-45}: same constructor arguments and state_dict keys (``act_emb.{0,2}``, ``fcs.{i}.model.{0,1}``, ``time_mlp``,
``mid_layer``, ``affine_in``, ``ln_resnet.{i}.net.{1,2,4}``, ``affine_out`` ...).

These modules are the parameter containers and the PyTorch (CPU / autograd) path.  The quirks the sampler must keep:
PearceMlp feeds the *raw* timestep as an input feature and divides skip paths by the literal 1.414;
a missing condition means a zero condition.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from ..utils import GroupNorm1d, Mlp
from .base_nn_diffusion import BaseNNDiffusion


# ------------------------------------------------------------------------------------------------ #
# PearceMlp                                                                                          #
# ------------------------------------------------------------------------------------------------ #
class TimeSiren(nn.Module):
    def __init__(self, input_dim, emb_dim):
        super().__init__()
        self.lin1 = nn.Linear(input_dim, emb_dim, bias=False)
        self.lin2 = nn.Linear(emb_dim, emb_dim)

    def forward(self, x):
        return self.lin2(torch.sin(self.lin1(x)))


class FCBlock(nn.Module):
    """Linear -> GroupNorm1d(8 groups, >=4 channels each) -> GELU(erf)."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(in_feats, out_feats), GroupNorm1d(out_feats, 8, 4), nn.GELU())

    def forward(self, x):
        return self.model(x)


class PearceMlp(BaseNNDiffusion):
    SKIP_SCALE = 1.414          # the reference's literal, not sqrt(2)

    def __init__(self, act_dim: int, To: int = 1, timestep_emb_type: str = "positional", emb_dim: int = 128,
                 hidden_dim: int = 512, timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.act_emb = nn.Sequential(nn.Linear(act_dim, emb_dim), nn.LeakyReLU(), nn.Linear(emb_dim, emb_dim))
        skip_in = hidden_dim + act_dim + 1
        self.fcs = nn.ModuleList([FCBlock(emb_dim * (2 + To), hidden_dim), FCBlock(skip_in, hidden_dim),
                                  FCBlock(skip_in, hidden_dim), nn.Linear(skip_in, act_dim)])
        self.To, self.emb_dim, self.act_dim, self.hidden_dim = To, emb_dim, act_dim, hidden_dim

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, act_dim), noise (b,), condition (b, To, emb_dim) | (b, To*emb_dim) | None -> (b, act_dim)."""
        from ..engine import train
        if train.supports_pearce(self, x, condition):
            return train.pearce_forward(self, x, noise, condition)          # autograd on, ROCm device: loss() / update()
        if condition is None:
            condition = torch.zeros(x.shape[0], self.To, self.emb_dim).to(x.device)
        t = noise.unsqueeze(-1)
        h = self.fcs[0](torch.cat([self.act_emb(x), self.map_noise(noise), torch.flatten(condition, 1)], -1))
        for block in (self.fcs[1], self.fcs[2]):
            skip = h / self.SKIP_SCALE
            h = block(torch.cat([skip, x, t], -1)) + skip
        return self.fcs[3](torch.cat([h, x, t], -1))


# ------------------------------------------------------------------------------------------------ #
# DQLMlp / IDQLMlp                                                                                    #
# ------------------------------------------------------------------------------------------------ #
def _time_mlp(emb_dim):
    return nn.Sequential(nn.Linear(emb_dim, emb_dim * 2), nn.Mish(), nn.Linear(emb_dim * 2, emb_dim))


class _ObsConditionedMlp(BaseNNDiffusion):
    """Shared front end: features = [x, time_mlp(map_noise(t)), condition-or-zeros]."""

    def _features(self, x, noise, condition):
        if condition is None:
            condition = torch.zeros(x.shape[0], self.obs_dim).to(x.device)
        return torch.cat([x, self.time_mlp(self.map_noise(noise)), condition], -1)


class DQLMlp(_ObsConditionedMlp):
    def __init__(self, obs_dim: int, act_dim: int, emb_dim: int = 16, timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.obs_dim = obs_dim
        self.time_mlp = _time_mlp(emb_dim)
        self.mid_layer = nn.Sequential(nn.Linear(obs_dim + act_dim + emb_dim, 256), nn.Mish(),
                                       nn.Linear(256, 256), nn.Mish(), nn.Linear(256, 256), nn.Mish())
        self.final_layer = nn.Linear(256, act_dim)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        from ..engine import train
        if train.supports_mlp(self, x, condition):       # autograd on, ROCm device (DQL back-propagates through sample()): Linear / Mish nodes on the library
            return train.dql_forward(self, x, noise, condition)
        return self.final_layer(self.mid_layer(self._features(x, noise, condition)))


class DVInvMlp(_ObsConditionedMlp):
    """Decision-Veteran inverse-dynamics denoiser (reference nn_diffusion/dvinvmlp.py:9-47): DQLMlp's trunk with a configurable
    width over features [x | time_mlp(map_noise(t)) | (obs, next_obs)]; the condition is REQUIRED (the reference concatenates it
    unconditionally).  Same parameter names as DQLMlp, so the batch-tiled MLP program serves it on a ROCm device."""

    def __init__(self, obs_dim: int, act_dim: int, emb_dim: int = 16, hidden_dim: int = 256,
                 timestep_emb_type: str = "positional", timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.obs_dim = obs_dim * 2
        self.time_mlp = _time_mlp(emb_dim)
        self.mid_layer = nn.Sequential(nn.Linear(obs_dim * 2 + act_dim + emb_dim, hidden_dim), nn.Mish(),
                                       nn.Linear(hidden_dim, hidden_dim), nn.Mish(), nn.Linear(hidden_dim, hidden_dim), nn.Mish())
        self.final_layer = nn.Linear(hidden_dim, act_dim)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: torch.Tensor = None):
        if condition is None:
            raise TypeError("DVInvMlp needs the (obs, next_obs) condition")       # reference: torch.cat fails on None
        from ..engine import train
        if train.supports_mlp(self, x, condition):
            return train.dql_forward(self, x, noise, condition)
        return self.final_layer(self.mid_layer(self._features(x, noise, condition)))


class ResidualBlock(nn.Module):
    """x + Linear(Mish(Linear(LayerNorm(Dropout(x)))))  (pre-norm MLP block, 4x expansion)."""

    def __init__(self, hidden_dim: int, dropout: float = 0.1):
        super().__init__()
        self.net = nn.Sequential(nn.Dropout(dropout), nn.LayerNorm(hidden_dim), nn.Linear(hidden_dim, hidden_dim * 4),
                                 nn.Mish(), nn.Linear(hidden_dim * 4, hidden_dim))

    def forward(self, x):
        return x + self.net(x)


class IDQLMlp(_ObsConditionedMlp):
    def __init__(self, obs_dim: int, act_dim: int, emb_dim: int = 64, hidden_dim: int = 256, n_blocks: int = 3,
                 dropout: float = 0.1, timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.obs_dim = obs_dim
        self.time_mlp = _time_mlp(emb_dim)
        self.affine_in = nn.Linear(obs_dim + act_dim + emb_dim, hidden_dim)
        self.ln_resnet = nn.Sequential(*[ResidualBlock(hidden_dim, dropout) for _ in range(n_blocks)])
        self.affine_out = self._make_head(hidden_dim, act_dim)

    @staticmethod
    def _make_head(hidden_dim, act_dim):
        return nn.Linear(hidden_dim, act_dim)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        from ..engine import dispatch, train
        if train.supports_idql(self, x, condition):   # autograd on, ROCm device (loss() / update()): Linear / LayerNorm nodes on the library
            return train.idql_forward(self, x, noise, condition)
        y = dispatch.try_backbone_forward(self, x, noise, condition)         # cdx_resmlp_run on a ROCm device
        if y is not None:
            return y
        return self._forward_torch(x, noise, condition)

    def _forward_torch(self, x, noise, condition=None):
        return self.affine_out(self.ln_resnet(self.affine_in(self._features(x, noise, condition))))


class NewIDQLMlp(IDQLMlp):
    """Same trunk, Mish before the output projection (``affine_out.1``)."""

    @staticmethod
    def _make_head(hidden_dim, act_dim):
        return nn.Sequential(nn.Mish(), nn.Linear(hidden_dim, act_dim))


# ------------------------------------------------------------------------------------------------ #
# MlpNNDiffusion                                                                                      #
# ------------------------------------------------------------------------------------------------ #
class MlpNNDiffusion(BaseNNDiffusion):
    def __init__(self, x_dim: int, emb_dim: int = 16, hidden_dims: List[int] = (256, 256),
                 activation: nn.Module = nn.ReLU(), timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.mlp = Mlp(x_dim + emb_dim, hidden_dims, x_dim, activation)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        t = self.map_noise(noise)
        t = t + (condition if condition is not None else torch.zeros_like(t))
        return self.mlp(torch.cat([x, t], -1))
