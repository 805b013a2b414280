"""Flux and source operators with the signatures of GaPFlow/integrate.py, evaluated on the GPU.

    predictor_corrector(q, p, tau, direction) -> (flux_x, flux_y)     integrate.py:38-77
    source(q, h, stress, stress_lower, stress_upper) -> out            integrate.py:80-130

NumPy arrays in, NumPy arrays out; each call is one H2D copy, one HIP kernel, one D2H copy.
(The time loop does not go through these wrappers -- the fused step kernel evaluates the same
expressions in registers -- they exist so code written against the reference's module keeps working.)
"""
import numpy as np

from . import _lib


def predictor_corrector(q, p, tau, direction):
    lib = _lib.require_device()
    q, p, tau = _lib.f64c(q), _lib.f64c(p), _lib.f64c(tau)
    if q.ndim != 3 or q.shape[0] != 3 or p.shape != q.shape[1:] or tau.shape != q.shape:
        raise ValueError("expected q (3,nx,ny), p (nx,ny), tau (3,nx,ny)")
    fx, fy = np.empty_like(q), np.empty_like(q)
    _lib.check(lib.gpf_predictor_corrector(q.shape[1], q.shape[2], _lib.as_dp(q), _lib.as_dp(p), _lib.as_dp(tau),
                                           int(direction), _lib.as_dp(fx), _lib.as_dp(fy)))
    return fx, fy


def source(q, h, stress, stress_lower, stress_upper):
    lib = _lib.require_device()
    q, stress = _lib.f64c(q), _lib.f64c(stress)
    h3 = _lib.f64c(np.asarray(h)[:3])           # the 4th slot (deformation) is unused, integrate.py:120-128
    lo, up = _lib.f64c(stress_lower), _lib.f64c(stress_upper)
    if q.ndim != 3 or q.shape[0] != 3 or h3.shape != q.shape or stress.shape != q.shape \
            or lo.shape != (6,) + q.shape[1:] or up.shape != lo.shape:
        raise ValueError("expected q,h[:3],stress (3,nx,ny) and stress_lower/upper (6,nx,ny)")
    out = np.empty_like(q)
    _lib.check(lib.gpf_source(q.shape[1], q.shape[2], _lib.as_dp(q), _lib.as_dp(h3), _lib.as_dp(stress),
                              _lib.as_dp(lo), _lib.as_dp(up), _lib.as_dp(out)))
    return out
