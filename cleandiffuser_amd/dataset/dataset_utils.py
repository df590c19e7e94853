"""The two loader helpers every training pipeline imports next to the solver (reference dataset/dataset_utils.py:386-412).
The datasets themselves (D4RL / robomimic / zarr readers) are outside the sampling hot path and are not rebuilt here."""
from ..utils.misc import dict_apply, loop_dataloader  # noqa: F401


def loop_two_dataloaders(dl1, dl2):
    """Endless stream of (batch1, batch2) pairs; each epoch is as long as the shorter loader (zip semantics)."""
    while True:
        yield from zip(dl1, dl2)
