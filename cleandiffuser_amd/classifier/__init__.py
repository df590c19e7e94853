"""Classifier-guidance wrappers (reference classifier/): base protocol, cumulative-reward, MSE and QGPO energy classifiers."""
from .base import BaseClassifier
from .rew_classifiers import CumRewClassifier
from .mse_classifier import MSEClassifier
from .qgpo_classifier import QGPOClassifier

__all__ = ["BaseClassifier", "CumRewClassifier", "MSEClassifier", "QGPOClassifier"]
