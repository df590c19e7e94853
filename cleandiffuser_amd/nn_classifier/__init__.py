"""Classifier networks (reference nn_classifier/): plug-in base, the half U-Net / half DiT trajectory scorers, MLP scorers."""
from .base_nn_classifier import BaseNNClassifier
from .half_jannerunet import HalfJannerUNet1d
from .half_dit import HalfDiT1d
from .mlp import MLPNNClassifier, QGPONNClassifier

__all__ = ["BaseNNClassifier", "HalfJannerUNet1d", "HalfDiT1d", "MLPNNClassifier", "QGPONNClassifier"]
