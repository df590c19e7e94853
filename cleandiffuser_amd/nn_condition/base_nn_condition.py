"""Condition-encoder plug-in base (contract: reference nn_condition/base_nn_condition.py:7-57).

Condition encoders run ONCE per ``sample()`` / ``loss()`` call (reference diffusionsde.py:499), never
inside the denoising loop; user-defined ones are ordinary PyTorch modules on whatever device the solver owns, the
built-in Linear / MLP ones (mlp.py) route their matmuls through the HIP library while sampling.
``forward(condition, mask=None) -> (b, *cond_out_shape)``; in train mode a Bernoulli label-dropout
mask is drawn, in eval mode ``mask=None`` means "keep everything".
"""
import torch
import torch.nn as nn

from ..utils import at_least_ndim


def get_mask(mask, mask_shape: tuple, dropout: float, train: bool, device):
    if train:
        return (torch.rand(mask_shape, device=device) > dropout).float()
    return 1. if mask is None else mask


class BaseNNCondition(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        raise NotImplementedError


class IdentityCondition(BaseNNCondition):
    """condition * mask, nothing else.  (b, *shape) -> (b, *shape)."""

    def __init__(self, dropout: float = 0.25):
        super().__init__()
        self.dropout = dropout

    def _mask(self, mask, batch, device, ndim):
        return at_least_ndim(get_mask(mask, (batch,), self.dropout, self.training, device), ndim)

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        return condition * self._mask(mask, condition.shape[0], condition.device, condition.dim())
