"""gaussian_processes_amd -- MI355X-native drop-in for jhamrick/gaussian_processes.

    import gaussian_processes_amd as gp
    g = gp.GP(gp.GaussianKernel(1.0, 0.5), x, y, s=1.0)
    g.log_lh, g.mean(xo), g.cov(xo)

Same package surface as the reference's ``gp`` (gp/__init__.py:1-14): ``GP``,
``Kernel``, ``GaussianKernel``, ``PeriodicKernel`` and the ``ext`` sub-package
(``gaussian_c``, ``periodic_c``, ``gp_c``).  All arithmetic of the fit/predict
path runs in hand-written HIP (libgpx.so, include/gpx.h); there is no CPU
fallback -- importing works anywhere, computing needs the built library and a GPU.
"""
import logging

from . import ext
from .gp import GP
from .kernels import Kernel, GaussianKernel, PeriodicKernel
from . import kernels

__all__ = ["ext", "GP", "Kernel", "PeriodicKernel", "GaussianKernel"]

logger = logging.getLogger("gp")
logger.setLevel("INFO")
