"""Training-data side of the pipelines: the loader helpers and the D4RL datasets with HBM-resident buffers: the
MuJoCo sequence / transition classes (d4rl_mujoco_dataset.py) and, round 5, their siblings over the same padded-episode store
(episode_store.py): multi-horizon / Decision-Veteran sequences and the transitions of MuJoCo, kitchen, antmaze and maze2d."""
from .base_dataset import BaseDataset  # noqa: F401
from .d4rl_mujoco_dataset import D4RLMuJoCoDataset, D4RLMuJoCoTDDataset, ResidentLoader  # noqa: F401
from .episode_store import (D4RLAntmazeDataset, D4RLAntmazeTDDataset, D4RLKitchenDataset, D4RLKitchenTDDataset,  # noqa: F401
                            D4RLMaze2DTDDataset, DV_D4RLAntmazeSeqDataset, DV_D4RLKitchenSeqDataset, DV_D4RLMaze2DSeqDataset,
                            DV_D4RLMuJoCoSeqDataset, EpisodeStore, MultiHorizonD4RLAntmazeDataset, MultiHorizonD4RLKitchenDataset,
                            MultiHorizonD4RLMuJoCoDataset)
