"""Gap-height profiles, restated from GaPFlow/topography.py:38-255.  Test infrastructure only."""
import numpy as np


def midpoint_grid(grid):
    # topography.py:38-54 -- cell centres including one ghost cell per side
    Nx, Ny, Lx, Ly = grid['Nx'], grid['Ny'], grid['Lx'], grid['Ly']
    x = np.arange(-1, Nx + 1) / Nx * Lx + (Lx / Nx) / 2.
    y = np.arange(-1, Ny + 1) / Ny * Ly + (Ly / Ny) / 2.
    return np.meshgrid(x, y, indexing='ij')


def profile(xx, yy, grid, geo, rng=None):
    """Return h, dh/dx, dh/dy on the (Nx+2, Ny+2) grid, before any flip."""
    Lx, Ly = grid['Lx'], grid['Ly']
    t = geo['type']
    hy = np.zeros_like(xx)
    if t == 'journal':          # topography.py:57-74
        freq = 2. * np.pi / Lx
        if 'eps' in geo:
            shift = geo['CR'] / freq
            amp = geo['eps'] * shift
        else:
            amp = (geo['hmax'] - geo['hmin']) / 2.
            shift = (geo['hmax'] + geo['hmin']) / 2.
        return shift + amp * np.cos(freq * xx), -amp * freq * np.sin(freq * xx), hy
    if t == 'inclined':         # topography.py:77-88
        slope = (geo['hmin'] - geo['hmax']) / Lx
        return geo['hmax'] + slope * xx, np.ones_like(xx) * slope, hy
    if t == 'parabolic':        # topography.py:91-104
        pre = 4. / Lx**2 * (geo['hmax'] - geo['hmin'])
        return pre * (xx - Lx / 2.)**2 + geo['hmin'], 2 * pre * (xx - Lx / 2.), hy
    if t == 'cdc':              # topography.py:107-130
        h0, h1, b = geo['hmin'], geo['hmax'], geo['b']
        slope = (h1 - h0) / (Lx / 2 - 2 * b)
        conv = np.logical_and(xx >= b, xx < Lx / 2 - b)
        center = np.logical_and(xx >= Lx / 2 - b, xx < Lx / 2 + b)
        div = np.logical_and(xx >= Lx / 2 + b, xx < Lx - b)
        h = np.ones_like(xx) * h1
        h[conv] = h1 - slope * (xx[conv] - b)
        h[center] = h0
        h[div] = h0 + slope * (xx[div] - (Lx / 2 + b))
        hx = np.zeros_like(h)
        hx[conv] = -slope
        hx[div] = slope
        return h, hx, hy
    if t == 'asperity':         # topography.py:133-170
        h0, h1, num = geo['hmin'], geo['hmax'], geo['num']
        if num == 1:
            hmins = np.array([h0])
        else:
            std = (h1 - h0) / 2. / 2.57
            rng = np.random if rng is None else rng     # unseeded in the reference (topography.py:146)
            hmins = rng.normal(loc=h0 + (h1 - h0) / 2., scale=std, size=num**2)
        xid = (xx // (Lx / num)).astype(int)
        yid = (yy // (Ly / num)).astype(int)
        bx = np.pi / (Lx / num)
        by = np.pi / (Ly / num)
        h = np.ones_like(xx) * h1
        hx = np.zeros_like(h)
        hy = np.zeros_like(h)
        k = 0
        for i in range(num):
            for j in range(num):
                m = np.logical_and(xid == i, yid == j)
                hm = hmins[k]
                k += 1
                cx = np.mean(xx[m])
                cy = np.mean(yy[m])
                h[m] -= (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
                hx[m] += bx * (h1 - hm) * (np.sin(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
                hy[m] += by * (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.sin(by * (yy[m] - cy)))
        return h, hx, hy
    raise ValueError(t)


def build_topography(grid, geo, rng=None):
    """(4, Nx+2, Ny+2) array [h, dh/dx, dh/dy, deformation] + xx, yy (topography.py:199-255)."""
    xx, yy = midpoint_grid(grid)
    h, hx, hy = profile(xx, yy, grid, geo, rng)
    topo = np.zeros((4,) + xx.shape)
    if geo['flip']:             # topography.py:227-234: transpose and swap the slope slots
        topo[0], topo[2], topo[1] = h.T, hx.T, hy.T
    else:
        topo[0], topo[1], topo[2] = h, hx, hy
    return topo, xx, yy


def central_gradients(h, dx, dy):
    # topography.py:273-280 (np.gradient: 2nd-order central, one-sided at the edges)
    return np.gradient(h, axis=0) / dx, np.gradient(h, axis=1) / dy
