from .base_nn_classifier import BaseNNClassifier
from .half_jannerunet import HalfJannerUNet1d
