"""Module-path alias: reference utils/building_blocks.py (implementations in utils/blocks.py and utils/critics.py)."""
from .blocks import GroupNorm1d, Mlp  # noqa: F401
from .critics import DQLCritic, IDQLQNet, IDQLVNet, SoftLowerBound, SoftUpperBound, TwinQ, V  # noqa: F401
# (the reference's Decision-Veteran critic and transformer toolkit -- DVHorizonCritic, Transformer, ... -- are outside the sampling path
#   and are not mirrored: a pipeline that wants them imports them from an installed reference)
