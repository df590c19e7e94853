"""Module-path alias: reference nn_diffusion/dvinvmlp.py (implementation in mlp_backbones.py)."""
from .base_nn_diffusion import BaseNNDiffusion  # noqa: F401
from .mlp_backbones import DVInvMlp  # noqa: F401
