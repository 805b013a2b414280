"""`GaPFlow.models.pressure.eos_pressure` (pressure.py:35-76) as a device operator."""
import ctypes as C

import numpy as np

from .. import _lib


def _eos_call(density, prop, want_sound):
    lib = _lib.require_device()
    name = prop['EOS']
    if name not in _lib.EOS_IDS:
        raise KeyError(f"unknown equation of state '{name}'")
    par = np.zeros(8)
    for i, k in enumerate(_lib.EOS_KEYS[name]):         # absent keys take the function defaults (pressure.py:73-76)
        if k not in prop and k not in _lib.EOS_DEFAULTS[name]:
            raise TypeError(f"EOS '{name}' needs the property '{k}'")
        par[i] = prop.get(k, _lib.EOS_DEFAULTS[name].get(k))
    rho = _lib.f64c(np.asarray(density, float))
    out = np.empty(rho.size)
    p_ptr, c_ptr = (None, out.ctypes.data_as(C.c_void_p)) if want_sound else (out.ctypes.data_as(C.c_void_p), None)
    _lib.check(lib.gpf_eos(_lib.EOS_IDS[name], par.ctypes.data_as(C.c_void_p), rho.size, rho.ctypes.data_as(C.c_void_p),
                           p_ptr, c_ptr))
    return out.reshape(rho.shape)


def eos_pressure(density, prop):
    """Pressure field of a density field for the equation of state named in prop['EOS']."""
    return _eos_call(density, prop, False)
