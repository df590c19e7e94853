"""Inverse-dynamics heads (reference invdynamic/): protocol class + the MLP implementations."""
from .mlp import BasicInvDynamic, FancyMlpInvDynamic, MlpInvDynamic

__all__ = ["BasicInvDynamic", "MlpInvDynamic", "FancyMlpInvDynamic"]
