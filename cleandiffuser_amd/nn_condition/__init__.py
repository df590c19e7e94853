from .base_nn_condition import BaseNNCondition, IdentityCondition, get_mask
from .mlp import (LinearCondition, MLPCondition, MLPSieveObsCondition, PearceObsCondition,
                  FourierCondition, PositionalCondition)
