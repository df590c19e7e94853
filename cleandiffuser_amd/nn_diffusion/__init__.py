from .base_nn_diffusion import BaseNNDiffusion
from .jannerunet import JannerUNet1d
from .mlp_backbones import PearceMlp, DQLMlp, IDQLMlp, NewIDQLMlp, MlpNNDiffusion
from .chiunet import ChiUNet1d
from .dit import DiT1d, DiT1Ref
from .chitransformer import ChiTransformer
