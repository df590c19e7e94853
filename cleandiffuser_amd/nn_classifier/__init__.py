from .base_nn_classifier import BaseNNClassifier
from .half_jannerunet import HalfJannerUNet1d
from .half_dit import HalfDiT1d
from .mlp import MLPNNClassifier, QGPONNClassifier
