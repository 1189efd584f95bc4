"""HIP-backed stand-ins for the reference's three Cython modules (gp/ext/__init__.py:1-5)."""
from . import gaussian_c
from . import periodic_c
from . import gp_c

__all__ = ["gaussian_c", "periodic_c", "gp_c"]
