"""Module-path alias: reference diffusion/vpode.py (implementation in edm_variants.py)."""
from .edm_variants import VPODE  # noqa: F401
