"""Module-path alias: reference utils/building_blocks.py (implementations in utils/blocks.py and utils/critics.py)."""
from .blocks import GroupNorm1d, Mlp  # noqa: F401
from .critics import (DQLCritic, DVHorizonCritic, DVTransformerBlock, FeedForward, IDQLQNet, IDQLVNet,  # noqa: F401
                      MultiHeadAttention, PreNorm, Residual, SoftLowerBound, SoftUpperBound, Transformer, TwinQ, V,
                      generate_causal_mask)
