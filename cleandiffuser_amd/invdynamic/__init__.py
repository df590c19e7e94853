from .common import BasicInvDynamic
from .mlp import MlpInvDynamic, FancyMlpInvDynamic
