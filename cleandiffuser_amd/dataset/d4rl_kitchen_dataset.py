"""Import-path alias of the reference module cleandiffuser/dataset/d4rl_kitchen_dataset.py: the classes live in episode_store.py."""
from .episode_store import D4RLKitchenDataset, D4RLKitchenTDDataset, DV_D4RLKitchenSeqDataset, MultiHorizonD4RLKitchenDataset  # noqa: F401
