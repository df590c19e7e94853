from .base import BaseClassifier
from .rew_classifiers import CumRewClassifier
from .mse_classifier import MSEClassifier
from .qgpo_classifier import QGPOClassifier
