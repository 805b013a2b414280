"""Elastic deformation of the gap under the film pressure (GaPFlow/topography.py:257-271, 327-437).

The reference hands the half-space response to ContactMechanics (FFT-based Green's functions); here the Green's function
is assembled once on the host, its Fourier transform lives on the device, and every time step the convolution
p -> u runs there (hipFFT real-to-complex transforms + small kernels, `gpf_elastic_update`), followed by the
under-relaxation, h = h_undeformed + u and the np.gradient stencil for dh/dx, dh/dy.

PARITY UNPINNED against ContactMechanics (not installed here, no fixture or test in the reference): the three response
functions follow the published forms -- periodic: u(q) = 2 p(q) / (E* |q|) with the q = 0 mode removed (Johnson,
Greenwood & Higginson 1985); free: Love's uniformly loaded rectangle on a doubled grid (Johnson 1985, eq. 3.25; Hockney &
Eastwood); semi-periodic: the free response summed over `n_images` periodic images on either side -- and the conventions of
the reference's wrapper (grid incl. ghost cells on (Lx, Ly), forces p dx dy, positive pressure -> positive displacement,
reference point [0, 0] subtracted unless fully periodic).  tests/test_oracle_elastic.py pins the same forms to analytic
solutions."""
import ctypes as C
import warnings

import numpy as np

from . import _lib


def _love(x, y, a, b, young):
    def term(p, q1, q2):
        return p * np.log((q1 + np.sqrt(q1 * q1 + p * p)) / (q2 + np.sqrt(q2 * q2 + p * p)))
    return (term(x + a, y + b, y - b) + term(y + b, x + a, x - a)
            + term(x - a, y - b, y + b) + term(y - b, x - a, x + a)) / (np.pi * young)


class ElasticDeformation:
    """Host half: which half-space applies, its Green's function in Fourier space, and the hand-over to the device."""

    def __init__(self, E, v, alpha_underrelax, grid, n_images):
        self.area_per_cell = grid['dx'] * grid['dy']
        nx, ny = grid['Nx'] + 2, grid['Ny'] + 2
        self.alpha_underrelax = alpha_underrelax
        perX, perY = bool(grid['bc_xE_P'][0]), bool(grid['bc_yS_P'][0])
        young_effective = E / (1 - v**2)
        Lx, Ly = grid['Lx'], grid['Ly']
        if (perX != perY) and ((perY and grid['Ny'] == 1) or (perX and grid['Nx'] == 1)):
            warnings.warn("You specified a semi-periodic 1D problem.\n"
                          "For the calculation of elastic deformation, we assume a line contact with "
                          "non-periodic boundary conditions in both directions.\n"
                          "For the calculation of the effective force F=p*A per cell, "
                          "we assume a unit length of {} = 1.".format("Ly" if perY else "Lx"))
            if perY:
                Ly = 1.0
            else:
                Lx = 1.0
            n_images = 0
        sx, sy = Lx / nx, Ly / ny
        self.area_per_pt = sx * sy
        if perX and perY:
            self.periodicity = 'full'
            self.shape_fft = (nx, ny)
            q = np.hypot(np.fft.fftfreq(nx, d=sx)[:, None], np.fft.rfftfreq(ny, d=sy)[None, :])
            with np.errstate(divide='ignore'):
                self.greens = np.where(q > 0, 1.0 / (np.pi * young_effective * q), 0.0).astype(complex)
            self._G_real = None
        else:
            self.periodicity = 'half' if perX != perY else 'none'
            if self.periodicity == 'none':
                n_images = 0
            px, py = (nx if perX else 2 * nx), (ny if perY else 2 * ny)
            self.shape_fft = (px, py)
            ix, iy = np.arange(px), np.arange(py)
            xs = (np.where(ix <= nx // 2, ix, ix - nx) if perX else np.where(ix <= nx, ix, ix - 2 * nx)) * sx
            ys = (np.where(iy <= ny // 2, iy, iy - ny) if perY else np.where(iy <= ny, iy, iy - 2 * ny)) * sy
            G = np.zeros((px, py))
            for k in range(-n_images, n_images + 1):
                G += _love(xs[:, None] + (k * Lx if perX else 0.0), ys[None, :] + (k * Ly if perY else 0.0),
                           sx / 2, sy / 2, young_effective)
            self._G_real = G
            self.greens = np.fft.rfft2(G)

    def attach(self, problem):
        """Upload the Green's function; from now on gpf_elastic_update(problem) deforms the gap on the device."""
        g = np.ascontiguousarray(np.stack([self.greens.real, self.greens.imag], axis=-1), dtype=np.float64)
        px, py = self.shape_fft
        _lib.check(problem._lib.gpf_elastic_setup(problem._h, px, py, g.ctypes.data_as(C.c_void_p), g.size,
                                                  float(self.alpha_underrelax), float(self.area_per_cell / self.area_per_pt),
                                                  0 if self.periodicity == 'full' else 1))

    # -- analysis helpers of the reference (topography.py:439-465) -----------------------------
    def get_G_real(self):
        if self._G_real is None:
            raise NotImplementedError("the periodic half-space is defined in Fourier space only")
        return np.fft.fftshift(self._G_real)

    def get_G_real_slices(self):
        G = self.get_G_real()
        return G[:, G.shape[1] // 2], G[G.shape[0] // 2, :]
