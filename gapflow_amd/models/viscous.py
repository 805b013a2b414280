"""`GaPFlow.models.viscous` as device operators: stress_bottom / stress_top / stress_avg (viscous.py:37, 281, 612).

Same signatures, NumPy in and out; the arithmetic runs in `gpf_viscous_stress` (csrc/closures.hpp `viscous_general`:
the parabolic velocity profile with Navier slip, differentiated in closed form).  slip="top" is the branch the solver
uses; any other keyword takes the reference's second branch (slip at both walls; Ls = 0 there means no slip)."""
import ctypes as C

import numpy as np

from .. import _lib


def _call(which, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip):
    lib = _lib.require_device()
    q, h = np.asarray(q, float), np.asarray(h, float)
    shape = np.broadcast_shapes(q.shape[1:], h.shape[1:], np.shape(Ls), np.shape(eta))
    n = int(np.prod(shape)) if shape else 1

    def comp3(a):
        a = np.asarray(a, float)[:3]
        return _lib.f64c(np.broadcast_to(a, (3,) + shape).reshape(3, n))

    def per_point(a):
        return _lib.f64c(np.broadcast_to(np.asarray(a, float), shape).reshape(n))

    qd, hd, etad, lsd = comp3(q), comp3(h), per_point(eta), per_point(Ls)
    gx = None if dqx is None else comp3(dqx)
    gy = None if dqy is None else comp3(dqy)
    ncomp = 3 if which == 2 else 6
    out = np.empty((ncomp, n))
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    outs = [None, None, None]
    outs[which] = out.ctypes.data_as(C.c_void_p)
    _lib.check(lib.gpf_viscous_stress(n, ptr(qd), ptr(hd), ptr(gx), ptr(gy), ptr(etad), ptr(lsd), float(U), float(V),
                                      float(zeta), 0 if slip == "top" else 1, *outs))
    return out.reshape((ncomp,) + shape)


def stress_bottom(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Viscous stress at the lower wall, Voigt order xx, yy, zz, yz, xz, xy (viscous.py:37-278)."""
    return _call(0, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)


def stress_top(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Viscous stress at the upper wall (viscous.py:281-609)."""
    return _call(1, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)


def stress_avg(q, h, U, V, eta, zeta, Ls, dqx=None, dqy=None, slip="top"):
    """Gap-averaged viscous stress xx, yy, xy (viscous.py:612-786).  The reference has branches for "top" and "both"
    only and returns its zero-initialised array for any other keyword (viscous.py:663, 717); so does this."""
    if slip not in ("top", "both"):
        q, h = np.asarray(q, float), np.asarray(h, float)
        return np.zeros((3,) + np.broadcast_shapes(q.shape[1:], h.shape[1:], np.shape(Ls), np.shape(eta)))
    return _call(2, q, h, U, V, eta, zeta, Ls, dqx, dqy, slip)
