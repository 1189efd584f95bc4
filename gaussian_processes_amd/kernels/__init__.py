"""Kernel plugin API (mirrors gp/kernels/__init__.py:1-5)."""
from .base import Kernel
from .periodic import PeriodicKernel
from .gaussian import GaussianKernel

__all__ = ["Kernel", "PeriodicKernel", "GaussianKernel"]
