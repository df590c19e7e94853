"""Module-path alias: reference diffusion/edmddim.py (implementation in edm_variants.py)."""
from .edm_variants import EDMDDIM  # noqa: F401
