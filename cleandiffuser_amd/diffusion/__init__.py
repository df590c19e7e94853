from .basic import DiffusionModel
from .diffusionsde import BaseDiffusionSDE, DiscreteDiffusionSDE, ContinuousDiffusionSDE, SUPPORTED_SOLVERS
from .newedm import ContinuousEDM
from .rectifiedflow import DiscreteRectifiedFlow, ContinuousRectifiedFlow
from .consistency_model import ContinuousConsistencyModel
from . import ddpm, dpmsolver, edm  # noqa: F401  legacy module paths: cleandiffuser.diffusion.{ddpm.DDPM, dpmsolver.DPMSolver, edm.EDM}
from .ddpm import DDPM                      # noqa: E402,F401  convenience exports (some pipelines import these from the package)
from .dpmsolver import DPMSolver            # noqa: E402,F401
from .edm import EDM                        # noqa: E402,F401
