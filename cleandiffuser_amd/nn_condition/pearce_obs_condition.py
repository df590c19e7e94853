"""Module-path alias: reference nn_condition/pearce_obs_condition.py (implementation in nn_condition/mlp.py)."""
from .base_nn_condition import IdentityCondition, get_mask  # noqa: F401
from .mlp import PearceObsCondition  # noqa: F401
