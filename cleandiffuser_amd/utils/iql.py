"""Module-path alias: reference utils/iql.py (implementation in utils/critics.py)."""
from .critics import IQL, TwinQ, V  # noqa: F401
