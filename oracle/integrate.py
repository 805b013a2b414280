"""Flux and source assembly, restated from GaPFlow/integrate.py:38-198.  Test infrastructure only."""
import numpy as np


def predictor_corrector(q, p, tau, direction):
    """One-sided flux differences (integrate.py:38-77).

    F_x = (jx, p + tau_xx, tau_xy), F_y = (jy, tau_xy, p + tau_yy) (integrate.py:133-198);
    direction=+1 -> F[i]-F[i-1], direction=-1 -> F[i+1]-F[i]; np.roll wraps over the
    whole array including ghost cells (integrate.py:74-75).
    """
    Fx = np.stack([q[1], p + tau[0], tau[2]])
    Fy = np.stack([q[2], tau[2], p + tau[1]])
    fx = -direction * (np.roll(Fx, direction, axis=1) - Fx)
    fy = -direction * (np.roll(Fy, direction, axis=2) - Fy)
    return fx, fy


def source(q, h, stress, lower, upper):
    """Source term, integrate.py:80-130 (Voigt order xx,yy,zz,yz,xz,xy for the wall arrays)."""
    out = np.zeros_like(q)
    out[0] = (-q[1] * h[1] - q[2] * h[2]) / h[0]
    out[1] = ((stress[0] - upper[0]) * h[1] + (stress[2] - upper[5]) * h[2] + upper[4] - lower[4]) / h[0]
    out[2] = ((stress[2] - upper[5]) * h[1] + (stress[1] - upper[1]) * h[2] + upper[3] - lower[3]) / h[0]
    return out
