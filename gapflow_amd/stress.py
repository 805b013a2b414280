"""Read-only views of the constitutive fields, with the member names of GaPFlow/models/stress.py.

In the reference these objects *own* the closure evaluation (Pressure.update, WallStress.update,
BulkStress.update; stress.py:289-362, 427-459, 600-622).  Here the closures are evaluated inside the
fused step kernel and never stored; the objects below materialise the fields on demand from the
current ``q`` (libgapflow_hip: gpf_update_closures + gpf_download) for diagnostics, output and tests.
"""
import numpy as np

from . import _lib


class _Model:
    is_gp_model = False
    use_active_learning = False

    def __init__(self, problem, surrogate=None):
        self._problem = problem
        self._surrogate = surrogate
        self.geo = problem.geo
        self.prop = problem.prop
        if surrogate is not None:
            self.is_gp_model = True
            self.use_active_learning = surrogate.use_active_learning

    def __getattr__(self, name):
        # GP members (variance, _infer_mean_var, history, kernel_lengthscale, ...) live on the surrogate
        s = self.__dict__.get('_surrogate')
        if s is not None and not name.startswith('__'):
            return getattr(s, name)
        raise AttributeError(name)


class Pressure(_Model):
    name = "zz"

    @property
    def pressure(self):
        """p(rho) on the full grid incl. ghost cells (stress.py:512-515, 622)."""
        return self._problem._derived(_lib.FIELD_PRESSURE)[0]

    @property
    def v_sound(self):
        """max over all cells of c(rho) (stress.py:522-539)."""
        return np.float64(self._problem._scalars().v_sound)


class BulkStress(_Model):
    name = "bulk"

    @property
    def stress(self):
        """Gap-averaged viscous stress xx, yy, xy (stress.py:407-410, 452-459)."""
        return self._problem._derived(_lib.FIELD_TAU_AVG)


class WallStress(_Model):
    """Wall stress object for direction 'x' (xz) or 'y' (yz).

    Each of the reference's two objects stores *half* of the shared components xx, yy, zz, xy and
    the full shear component of its own direction (index 4 for 'x', 3 for 'y'), so that the sum
    of both objects is the full tensor (stress.py:346-362, problem.py:554-555).
    """

    def __init__(self, problem, direction='x', surrogate=None):
        super().__init__(problem, surrogate)
        self.name = f'{direction}z'
        self._out_index = {'x': 4, 'y': 3}[direction]

    def _split(self, full):
        out = np.zeros_like(full)
        for k in (0, 1, 2, 5):
            out[k] = full[k] / 2.
        out[self._out_index] = full[self._out_index]
        return out

    @property
    def lower(self):
        return self._split(self._problem._derived(_lib.FIELD_WALL_LOWER))

    @property
    def upper(self):
        return self._split(self._problem._derived(_lib.FIELD_WALL_UPPER))

    @property
    def full(self):
        return np.concatenate([self.lower, self.upper])
