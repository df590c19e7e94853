"""Cumulative-reward regressor used as Diffuser's guidance / candidate-ranking signal
(reference classifier/rew_classifiers.py:7-29): ``logp(x, t) = model_ema(x, t)``; MSE training target R.

The training step is the base class's (no gradient clipping is configured, so it is a plain Adam step followed by the EMA update);
only the returned dict is narrowed to what the reference reports.  ``logp`` / ``gradients`` on a ROCm device are served by the
explicit forward+backward kernels of engine/classifier_grad.py when the network is a HalfJannerUNet1d."""
from typing import Optional

from .base import BaseClassifier


class CumRewClassifier(BaseClassifier):
    def __init__(self, nn_classifier, device: str = "cpu", optim_params: Optional[dict] = None):
        super().__init__(nn_classifier, ema_rate=0.995, grad_clip_norm=None, optim_params=optim_params, device=device)

    def loss(self, x, noise, R):
        err = self.model(x, noise, None) - R
        return (err * err).mean()

    def update(self, x, noise, R):
        return {"loss": super().update(x, noise, R)["loss"]}

    def logp(self, x, noise, c=None):
        return self.model_ema(x, noise)
