"""Dataset protocol of the training pipelines (reference cleandiffuser/dataset/base_dataset.py:24-44): a ``torch.utils.data.Dataset``
whose items are dictionaries {"obs": {key: (T, Do)}, "act": (T, Da), ...} plus ``get_normalizer()``."""
from typing import Dict

import torch
from torch.utils.data import Dataset


class BaseDataset(Dataset):
    def get_normalizer(self, **kwargs):
        raise NotImplementedError()

    def __len__(self) -> int:
        return 0

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()
