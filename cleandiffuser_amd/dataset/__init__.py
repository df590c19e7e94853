"""Training-data side of the pipelines: the loader helpers and the D4RL-MuJoCo datasets with HBM-resident buffers."""
from .base_dataset import BaseDataset  # noqa: F401
from .d4rl_mujoco_dataset import D4RLMuJoCoDataset, D4RLMuJoCoTDDataset, ResidentLoader  # noqa: F401
