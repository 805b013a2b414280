"""`GaPFlow.models.viscosity` as device operators: piezoviscosity, shear_thinning_factor, shear_rate_avg
(viscosity.py:34-141); NumPy in and out, arithmetic in `gpf_viscosity`.  Absent parameters take the defaults of the
reference's law functions; an unknown law name leaves the viscosity unchanged, as there (viscosity.py:63-64, 93-94)."""
import ctypes as C

import numpy as np

from .. import _lib


def _run(kind, law, par, mu0, arrays, u1=0., u2=0.):
    lib = _lib.require_device()
    shape = np.broadcast_shapes(*[np.shape(a) for a in arrays])
    flat = [_lib.f64c(np.broadcast_to(np.asarray(a, float), shape).reshape(-1)) for a in arrays]
    n = flat[0].size
    out = np.empty(n)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    p = np.zeros(4)
    p[:len(par)] = par
    args = [ptr(a) for a in flat] + [None] * (3 - len(flat))
    _lib.check(lib.gpf_viscosity(kind, law, ptr(p), float(mu0), n, *args, float(u1), float(u2), ptr(out)))
    return out.reshape(shape)


def piezoviscosity(p, mu0, piezo_dict):
    """Pressure- (or, for the Dukler / McAdams mixture laws, density-) dependent viscosity."""
    name = piezo_dict.get('name')
    if name not in _lib.PIEZO_IDS:
        return np.ones_like(np.asarray(p, float)) * mu0
    par = [piezo_dict.get(k, _lib.PIEZO_DEFAULTS[name][k]) for k in _lib.PIEZO_KEYS[name]]
    return _run(0, _lib.PIEZO_IDS[name], par, mu0, [p])


def shear_thinning_factor(shear_rate, mu0, thinning_dict):
    """mu(shear rate) / mu0."""
    name = thinning_dict.get('name')
    if name not in _lib.THINNING_IDS:
        return np.ones_like(np.asarray(shear_rate, float))
    par = [thinning_dict.get(k, _lib.THINNING_DEFAULTS[name][k]) for k in _lib.THINNING_KEYS[name]]
    return _run(1, _lib.THINNING_IDS[name], par, mu0, [shear_rate])


def shear_rate_avg(dp_dx, dp_dy, h, u1, u2, mu):
    """Mean of the absolute wall shear rates of the Newtonian profile."""
    return _run(2, 0, [], mu, [dp_dx, dp_dy, h], u1, u2)
