"""gapflow_amd -- MI355X-native (gfx950) implementation of GaPFlow's explicit time-integration hot path.

Drop-in surface (same names and semantics as the reference package ``GaPFlow``):

    from gapflow_amd import Problem
    Problem.from_yaml("input.yaml").run()

The arithmetic of every time step runs in hand-written HIP kernels (libgapflow_hip.so, C ABI in
include/gapflow_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

from .problem import Problem  # noqa: E402,F401
from .gp import Database  # noqa: E402,F401  (GaPFlow/__init__.py:36)
