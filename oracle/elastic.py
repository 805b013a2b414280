"""Elastic half-space deformation of the gap, restated from GaPFlow/topography.py:257-271, 327-437.  Test infrastructure
only.

PARITY UNPINNED.  The reference delegates the arithmetic to ContactMechanics>=1.8.0 (pyproject.toml), which is not
installed here and not vendored, and holds no test or fixture for this path.  What follows restates the published
algorithms that library implements, from memory of its FFTElasticHalfSpace module:

  * periodic half-space (Johnson, Greenwood & Higginson 1985, eq. A.2; Stanley & Kato 1997): in Fourier space
    u(q) = 2 p(q) / (E* |q|), the q = 0 mode set to zero (the reference passes stiffness_q0=0.0, topography.py:384);
  * free half-space (Love 1929 / Johnson 1985 eq. 3.25 for a uniformly loaded rectangle, Hockney & Eastwood's doubled
    grid for the aperiodic convolution);
  * semi-periodic: the free kernel summed over `n_images` periodic images on either side along the periodic direction,
    doubled grid along the other one.  The reference's class for this case (SemiPeriodicFFTElasticHalfSpace) is newer
    than what can be recalled here; this is the natural reading of its arguments (periodicity, n_images).

Conventions kept from the reference: the grid handed to the half-space includes the ghost cells, (Nx+2, Ny+2) points on
(Lx, Ly) (topography.py:357, 382-399), forces are p dx dy (topography.py:415), the library divides by ITS cell area
Lx Ly / ((Nx+2)(Ny+2)), and positive pressure gives positive displacement (topography.py:416).  The analytic checks in
tests/test_oracle_elastic.py (sinusoidal load, direct summation) pin the restatement to the physics."""
import numpy as np


def love_kernel(x, y, a, b, young):
    """Surface displacement at (x, y) of a unit pressure on the rectangle |x| <= a, |y| <= b (Johnson eq. 3.25)."""
    def term(p, q1, q2):
        # p * ln((q1 + sqrt(q1^2 + p^2)) / (q2 + sqrt(q2^2 + p^2)))
        return p * np.log((q1 + np.sqrt(q1 * q1 + p * p)) / (q2 + np.sqrt(q2 * q2 + p * p)))
    return (term(x + a, y + b, y - b) + term(y + b, x + a, x - a)
            + term(x - a, y - b, y + b) + term(y - b, x - a, x + a)) / (np.pi * young)


class ElasticDeformation:
    def __init__(self, E, v, alpha_underrelax, grid, n_images):
        self.area_per_cell = grid['dx'] * grid['dy']
        nx, ny = grid['Nx'] + 2, grid['Ny'] + 2
        self.n = (nx, ny)
        self.u_prev = np.zeros((nx, ny))
        self.alpha = alpha_underrelax
        perX, perY = bool(grid['bc_xE_P'][0]), bool(grid['bc_yS_P'][0])
        young = E / (1 - v**2)
        Lx, Ly = grid['Lx'], grid['Ly']
        if perX != perY and ((perY and grid['Ny'] == 1) or (perX and grid['Nx'] == 1)):     # topography.py:366-380
            if perY:
                Ly = 1.0
            else:
                Lx = 1.0
            n_images = 0
        sx, sy = Lx / nx, Ly / ny
        self.area_per_pt = sx * sy
        if perX and perY:
            self.periodicity = 'full'
            self.pad = (nx, ny)
            qx = np.fft.fftfreq(nx, d=sx)[:, None]
            qy = np.fft.rfftfreq(ny, d=sy)[None, :]
            q = np.sqrt(qx**2 + qy**2)                      # cycles per length
            with np.errstate(divide='ignore'):
                self.greens = np.where(q > 0, 1.0 / (np.pi * young * q), 0.0).astype(complex)
        else:
            self.periodicity = 'half' if perX != perY else 'none'
            px, py = (nx if perX else 2 * nx), (ny if perY else 2 * ny)
            self.pad = (px, py)
            ix, iy = np.arange(px), np.arange(py)
            # signed offsets: aperiodic direction on the doubled grid, periodic one wrapped at half the period
            xs = (np.where(ix <= nx, ix, ix - 2 * nx) if not perX else np.where(ix <= nx // 2, ix, ix - nx)) * sx
            ys = (np.where(iy <= ny, iy, iy - 2 * ny) if not perY else np.where(iy <= ny // 2, iy, iy - ny)) * sy
            X, Y = xs[:, None], ys[None, :]
            G = np.zeros((px, py))
            if self.periodicity == 'none':
                n_images = 0                                # nothing to repeat
            for k in range(-n_images, n_images + 1):
                G += love_kernel(X + (k * Lx if perX else 0.0), Y + (k * Ly if perY else 0.0), sx / 2, sy / 2, young)
            self.G_real = G
            self.greens = np.fft.rfft2(G)

    def get_deformation(self, p):
        nx, ny = self.n
        forces = np.zeros(self.pad)
        forces[:nx, :ny] = p * self.area_per_cell
        disp = np.fft.irfft2(self.greens * np.fft.rfft2(forces), s=self.pad)[:nx, :ny] / self.area_per_pt
        return disp

    def get_deformation_underrelax(self, p):
        u = (1 - self.alpha) * self.u_prev + self.alpha * self.get_deformation(p)
        self.u_prev = u.copy()
        return u

    def update(self, p):
        """Topography.update (topography.py:257-269): the deformation to add to the undeformed gap height."""
        if self.periodicity in ('half', 'none'):
            d = self.get_deformation_underrelax(p - p[0, 0])
            return d - d[0, 0]
        return self.get_deformation_underrelax(p)
