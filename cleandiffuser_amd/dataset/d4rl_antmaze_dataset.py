"""Import-path alias of the reference module cleandiffuser/dataset/d4rl_antmaze_dataset.py: the classes live in episode_store.py."""
from .episode_store import D4RLAntmazeDataset, D4RLAntmazeTDDataset, DV_D4RLAntmazeSeqDataset, MultiHorizonD4RLAntmazeDataset  # noqa: F401
