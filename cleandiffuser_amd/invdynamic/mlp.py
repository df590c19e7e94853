"""Inverse-dynamics heads a(o, o') applied to consecutive states of a sampled plan (Decision Diffuser & co.: reference
invdynamic/mlp.py:7-293; call site pipelines/dd_d4rl_antmaze.py:142-143).  Same constructor arguments, attribute names and
checkpoint keys as the reference (``mlp`` / ``model`` state dicts); same quirk: ``optim_params`` is what reaches Adam, the
per-class default learning rate is computed but unused (invdynamic/mlp.py:47-52).

``predict`` on a ROCm device runs the chain through engine/heads.py (fp32-MFMA GEMMs with fused bias + activation, fused
LayerNorm); ``update`` keeps the reference's autograd graph with every Linear / LayerNorm / activation node on the library's kernels
(engine/train.py:chain_forward) and steps through ``FusedAdam`` (``cdx_optim_f32``).
"""
import torch
import torch.nn as nn

from ..utils import Mlp


def _forward_rows(net: nn.Module, x: torch.Tensor) -> torch.Tensor:
    from ..engine import heads, train
    seq = net.mlp if isinstance(net, Mlp) else net
    if train.supports_chain(seq, x):           # autograd on, ROCm device (``update``): library nodes forward and backward
        return train.chain_forward(seq, x)
    y = heads.try_sequential(seq, x)
    return net(x) if y is None else y


def _adam(params, optim_params):
    """What the reference builds with ``torch.optim.Adam`` (invdynamic/mlp.py:47-52): the same class on CPU, its one-launch subclass on a
    ROCm device."""
    from ..engine.optim import FusedAdam
    return FusedAdam(params, **optim_params)


class BasicInvDynamic:
    """Protocol of an inverse-dynamics head (reference invdynamic/common.py:1-6): ``predict`` with keyword arguments, and calling the
    object is calling ``predict``."""

    def predict(self, **kwargs):
        raise NotImplementedError

    def __call__(self, **kwargs):
        return self.predict(**kwargs)


class _MseTrainedHead:
    """update/predict/train/eval/save/load shared by the heads; subclasses provide ``forward`` and ``_net``."""

    def update(self, o, a, o_next):
        self.optim.zero_grad()
        loss = ((self.forward(o, o_next) - a) ** 2).mean()
        loss.backward()
        self.optim.step()
        return {"loss": loss.item()}

    @torch.no_grad()
    def predict(self, o, o_next):
        return self.forward(o, o_next)

    def __call__(self, o, o_next):
        return self.predict(o, o_next)

    def train(self):
        self._net().train()

    def eval(self):
        self._net().eval()

    def save(self, path):
        torch.save(self._net().state_dict(), path)

    def load(self, path):
        self._net().load_state_dict(torch.load(path, self.device))


class MlpInvDynamic(_MseTrainedHead):
    """Linear(2o, h)-ReLU-Linear(h, h)-ReLU-Linear(h, a)-out_activation, orthogonal weights, zero biases."""

    def __init__(self, o_dim: int, a_dim: int, hidden_dim: int = 512, out_activation: nn.Module = nn.Tanh(),
                 optim_params: dict = {}, device: str = "cpu"):
        self.device = device
        self.o_dim, self.a_dim, self.hidden_dim = o_dim, a_dim, hidden_dim
        self.out_activation = out_activation
        self.optim_params = optim_params
        self.mlp = Mlp(2 * o_dim, [hidden_dim, hidden_dim], a_dim, nn.ReLU(), out_activation).to(device)
        self.optim = _adam(self.mlp.parameters(), optim_params)
        self._init_weights()

    def _net(self):
        return self.mlp

    def _init_weights(self):
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, o, o_next):
        return _forward_rows(self.mlp, torch.cat([o, o_next], dim=-1))


class FancyMlpInvDynamic(_MseTrainedHead):
    """Linear-GELU-[LayerNorm]-[Dropout 0.1]-Linear-GELU-Linear-out_activation (default torch init)."""

    def __init__(self, o_dim: int, a_dim: int, hidden_dim: int = 256, out_activation: nn.Module = nn.Tanh(),
                 add_norm: bool = False, add_dropout: bool = False, optim_params: dict = {}, device: str = "cpu"):
        self.device = device
        self.o_dim, self.a_dim, self.hidden_dim = o_dim, a_dim, hidden_dim
        self.out_activation = out_activation
        self.optim_params = optim_params
        self.model = nn.Sequential(
            nn.Linear(2 * o_dim, hidden_dim), nn.GELU(),
            nn.LayerNorm(hidden_dim) if add_norm else nn.Identity(),
            nn.Dropout(0.1) if add_dropout else nn.Identity(),
            nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, a_dim), out_activation).to(device)
        self.optim = _adam(self.model.parameters(), optim_params)

    def _net(self):
        return self.model

    def forward(self, o, o_next):
        return _forward_rows(self.model, torch.cat([o, o_next], dim=-1))


class EnsembleMlpInvDynamic(MlpInvDynamic):
    """n_models heads averaged at predict time; ``mlp_type`` "standard" (Mlp) or "fancy" (Linear-LN-Mish stacks)."""

    def __init__(self, o_dim: int, a_dim: int, hidden_dim: int = 512, out_activation: nn.Module = nn.Identity(),
                 optim_params: dict = {}, n_models=5, mlp_type="standard", device: str = "cpu"):
        assert mlp_type in ["standard", "fancy"]
        super().__init__(o_dim, a_dim, hidden_dim, out_activation, optim_params, device)
        self.n_models = n_models
        h = hidden_dim
        if mlp_type == "standard":
            members = [Mlp(2 * o_dim, [h, h], a_dim, nn.ReLU(), out_activation) for _ in range(n_models)]
        else:       # the last Linear maps to hidden_dim, not a_dim -- kept as the reference has it (invdynamic/mlp.py:189-193)
            members = [nn.Sequential(nn.Linear(2 * o_dim, h), nn.LayerNorm(h), nn.Mish(), nn.Dropout(0.1),
                                     nn.Linear(h, h), nn.LayerNorm(h), nn.Mish(), nn.Linear(h, h), out_activation)
                       for _ in range(n_models)]
        self.mlp = nn.ModuleList(members).to(device)
        self.optim = _adam(self.mlp.parameters(), self.optim_params)
        self._init_weights()

    def forward(self, o, o_next, idx=None):
        x = torch.cat([o, o_next], dim=-1)
        if idx is not None:
            return _forward_rows(self.mlp[idx], x)
        return sum(_forward_rows(m, x) for m in self.mlp) / self.n_models

    def update_idx(self, idx, o, a, o_next):
        self.optim.zero_grad()
        loss = ((self.forward(o, o_next, idx) - a) ** 2).mean()
        loss.backward()
        self.optim.step()
        return loss.item()


class ResidualBlock(nn.Module):
    def __init__(self, hidden_dim: int = 256, add_norm: bool = False, add_dropout: bool = False):
        super().__init__()
        self.norm = nn.LayerNorm(hidden_dim) if add_norm else nn.Identity()
        self.mlp = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                 nn.Dropout(0.1) if add_dropout else nn.Identity(), nn.Linear(hidden_dim, hidden_dim))

    def forward(self, x):
        x = self.norm(x)
        return x + self.mlp(x)


class ResInvDynamic(_MseTrainedHead):
    """pre_linear -> n_blocks x ResidualBlock -> post_linear (development head in the reference, invdynamic/mlp.py:232-293)."""

    def __init__(self, o_dim: int, a_dim: int, hidden_dim: int = 256, out_activation: nn.Module = nn.Tanh(),
                 add_norm: bool = False, add_dropout: bool = False, n_blocks: int = 1, optim_params: dict = {},
                 device: str = "cpu"):
        self.device = device
        self.n_blocks = n_blocks
        self.o_dim, self.a_dim, self.hidden_dim = o_dim, a_dim, hidden_dim
        self.out_activation = out_activation
        self.optim_params = optim_params
        self.model = nn.ModuleDict({
            "pre_linear": nn.Sequential(nn.Linear(2 * o_dim, hidden_dim), nn.GELU()).to(device),
            "post_linear": nn.Sequential(nn.Linear(hidden_dim, a_dim), out_activation).to(device)})
        for i in range(n_blocks):
            self.model[f"res_block{i}"] = ResidualBlock(hidden_dim, add_norm, add_dropout).to(device)
        self.optim = _adam(self.model.parameters(), optim_params)

    def _net(self):
        return self.model

    def forward(self, o, o_next):
        h = _forward_rows(self.model["pre_linear"], torch.cat([o, o_next], dim=-1))
        for i in range(self.n_blocks):
            blk = self.model[f"res_block{i}"]
            h = blk.norm(h)
            h = h + _forward_rows(blk.mlp, h)
        return _forward_rows(self.model["post_linear"], h)
