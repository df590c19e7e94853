"""Module-path alias: reference nn_diffusion/mlps.py (implementation in mlp_backbones.py)."""
from .base_nn_diffusion import BaseNNDiffusion  # noqa: F401
from .mlp_backbones import MlpNNDiffusion  # noqa: F401
