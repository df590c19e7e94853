"""Dataset-side normalisers (numpy; applied to observations before they become a sampling condition and inverted on the sampled
actions): reference utils/normalizers.py:8-127.  Statistics are taken over the leading ``start_dim`` axes of the dataset."""
from typing import Optional

import numpy as np

from .misc import at_least_ndim


def _leading_axes(X: np.ndarray, start_dim: int):
    return tuple(range(start_dim + X.ndim if start_dim < 0 else start_dim))


class EmptyNormalizer:
    """Identity."""

    def normalize(self, x: np.ndarray):
        return x

    def unnormalize(self, x: np.ndarray):
        return x


class GaussianNormalizer(EmptyNormalizer):
    """Zero mean / unit variance per feature; constant features keep std 1 (so they normalise to 0)."""

    def __init__(self, X: np.ndarray, start_dim: int = -1):
        axes = _leading_axes(X, start_dim)
        self.mean = np.mean(X, axis=axes)
        self.std = np.std(X, axis=axes)
        self.std[self.std == 0] = 1.

    def normalize(self, x: np.ndarray):
        return (x - at_least_ndim(self.mean, x.ndim, 1)) / at_least_ndim(self.std, x.ndim, 1)

    def unnormalize(self, x: np.ndarray):
        return x * at_least_ndim(self.std, x.ndim, 1) + at_least_ndim(self.mean, x.ndim, 1)


class MinMaxNormalizer(EmptyNormalizer):
    """[min, max] -> [-1, 1] per feature; zero-range features are masked to 0 in both directions."""

    def __init__(self, X: np.ndarray, start_dim: int = -1, X_max: Optional[np.ndarray] = None,
                 X_min: Optional[np.ndarray] = None):
        axes = _leading_axes(X, start_dim)
        self.max = np.max(X, axis=axes) if X_max is None else X_max
        self.min = np.min(X, axis=axes) if X_min is None else X_min
        self.mask = np.ones_like(self.max)
        self.range = self.max - self.min
        self.mask[self.max == self.min] = 0.
        self.range[self.range == 0] = 1.

    def normalize(self, x: np.ndarray):
        unit = (x - at_least_ndim(self.min, x.ndim, 1)) / at_least_ndim(self.range, x.ndim, 1)
        return (unit * 2 - 1) * at_least_ndim(self.mask, x.ndim, 1)

    def unnormalize(self, x: np.ndarray):
        unit = (x + 1) / 2 * at_least_ndim(self.mask, x.ndim, 1)
        return unit * at_least_ndim(self.range, x.ndim, 1) + at_least_ndim(self.min, x.ndim, 1)
