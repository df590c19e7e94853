"""Module-path alias: reference utils/typings.py."""
from .misc import TensorDict  # noqa: F401
