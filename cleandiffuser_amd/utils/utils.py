"""Module-path alias: reference utils/utils.py (schedules, embeddings and helpers live in schedules.py / embeddings.py / misc.py)."""
from .embeddings import *  # noqa: F401,F403
from .embeddings import SUPPORTED_TIMESTEP_EMBEDDING  # noqa: F401
from .misc import *  # noqa: F401,F403
from .misc import DD_RETURN_SCALE, TensorDict, param_to_module  # noqa: F401
from .schedules import *  # noqa: F401,F403
from .schedules import SUPPORTED_DISCRETIZATIONS, SUPPORTED_NOISE_SCHEDULES, SUPPORTED_SAMPLING_STEP_SCHEDULE  # noqa: F401
