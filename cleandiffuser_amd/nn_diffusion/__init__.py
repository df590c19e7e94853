from .base_nn_diffusion import BaseNNDiffusion
from .jannerunet import JannerUNet1d
from .mlp_backbones import PearceMlp, DQLMlp, DVInvMlp, IDQLMlp, NewIDQLMlp, MlpNNDiffusion
from .sfbc_unet import SfBCUNet
from .pearcetransformer import PearceTransformer
from .chiunet import ChiUNet1d
from .dit import DiT1d, DiT1Ref
from .chitransformer import ChiTransformer
