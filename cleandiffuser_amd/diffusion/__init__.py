from .basic import DiffusionModel
from .diffusionsde import BaseDiffusionSDE, DiscreteDiffusionSDE, ContinuousDiffusionSDE, SUPPORTED_SOLVERS
from .newedm import ContinuousEDM
