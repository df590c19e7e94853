"""Import-path alias of the reference module cleandiffuser/dataset/d4rl_maze2d_dataset.py: the classes live in episode_store.py."""
from .episode_store import D4RLMaze2DTDDataset, DV_D4RLMaze2DSeqDataset  # noqa: F401
