"""Module-path alias: reference invdynamic/common.py (the protocol class lives next to its implementations in mlp.py)."""
from .mlp import BasicInvDynamic  # noqa: F401
