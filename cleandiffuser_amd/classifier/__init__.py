from .base import BaseClassifier
from .rew_classifiers import CumRewClassifier
