"""Module-path alias: reference nn_condition/positional.py (implementation in nn_condition/mlp.py)."""
from .mlp import FourierCondition, MLPCondition, PositionalCondition  # noqa: F401
