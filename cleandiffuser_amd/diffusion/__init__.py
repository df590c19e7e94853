from .basic import DiffusionModel
from .diffusionsde import BaseDiffusionSDE, DiscreteDiffusionSDE, ContinuousDiffusionSDE, SUPPORTED_SOLVERS
from .newedm import ContinuousEDM
from .rectifiedflow import DiscreteRectifiedFlow, ContinuousRectifiedFlow
from .consistency_model import ContinuousConsistencyModel
from . import ddpm, dpmsolver, edm  # noqa: F401  legacy module paths: cleandiffuser.diffusion.{ddpm.DDPM, dpmsolver.DPMSolver, edm.EDM}
