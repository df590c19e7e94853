from .base_nn_diffusion import BaseNNDiffusion
from .jannerunet import JannerUNet1d
