"""Module-path alias: reference diffusion/veode.py (implementation in edm_variants.py)."""
from .edm_variants import VEODE  # noqa: F401
