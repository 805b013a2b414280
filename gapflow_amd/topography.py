"""Gap topography: analytic height profiles and their slopes on the cell-centre grid.

Host-side set-up code (runs once per problem); mirrors GaPFlow/topography.py:38-324.  The
elastic half-space coupling (topography.py:327-465) lives in gapflow_amd/elastic.py and on the device.
"""
import numpy as np


def create_midpoint_grid(disc):
    """Cell centres incl. one ghost cell per side: x_i = (i + 1/2) dx, i = -1..Nx (topography.py:38-54)."""
    Lx, Ly, Nx, Ny = disc['Lx'], disc['Ly'], disc['Nx'], disc['Ny']
    x = np.arange(-1, Nx + 1) / Nx * Lx + (Lx / Nx) / 2.
    y = np.arange(-1, Ny + 1) / Ny * Ly + (Ly / Ny) / 2.
    return np.meshgrid(x, y, indexing='ij')


def journal_bearing(xx, grid, geo):
    # topography.py:57-74
    freq = 2. * np.pi / grid['Lx']
    if 'eps' in geo.keys():
        shift = geo['CR'] / freq
        amp = geo['eps'] * shift
    else:
        amp = (geo['hmax'] - geo['hmin']) / 2.
        shift = (geo['hmax'] + geo['hmin']) / 2.
    return shift + amp * np.cos(freq * xx), -amp * freq * np.sin(freq * xx), np.zeros_like(xx)


def inclined_slider(xx, grid, geo):
    # topography.py:77-88
    slope = (geo['hmin'] - geo['hmax']) / grid['Lx']
    return geo['hmax'] + slope * xx, np.ones_like(xx) * slope, np.zeros_like(xx)


def parabolic_slider(xx, grid, geo):
    # topography.py:91-104
    Lx = grid['Lx']
    prefac = 4. / Lx**2 * (geo['hmax'] - geo['hmin'])
    return prefac * (xx - Lx / 2.)**2 + geo['hmin'], 2 * prefac * (xx - Lx / 2.), np.zeros_like(xx)


def cdc(xx, grid, geo):
    # topography.py:107-130 (converging - flat - diverging)
    Lx, h0, h1, b = grid['Lx'], geo['hmin'], geo['hmax'], geo['b']
    slope = (h1 - h0) / (Lx / 2 - 2 * b)
    conv = (xx >= b) & (xx < Lx / 2 - b)
    center = (xx >= Lx / 2 - b) & (xx < Lx / 2 + b)
    div = (xx >= Lx / 2 + b) & (xx < Lx - b)
    h = np.full_like(xx, h1)
    h[conv] = h1 - slope * (xx[conv] - b)
    h[center] = h0
    h[div] = h0 + slope * (xx[div] - (Lx / 2 + b))
    dh_dx = np.zeros_like(h)
    dh_dx[conv] = -slope
    dh_dx[div] = slope
    return h, dh_dx, np.zeros_like(h)


def asperity(xx, yy, grid, geo):
    # topography.py:133-170; for num > 1 the minimum heights are drawn from an unseeded
    # normal distribution, exactly as in the reference (topography.py:146)
    h0, h1, num = geo['hmin'], geo['hmax'], geo['num']
    Lx, Ly = grid['Lx'], grid['Ly']
    if num == 1:
        hmins = np.array([h0])
    else:
        std = (h1 - h0) / 2. / 2.57
        hmins = np.random.normal(loc=h0 + (h1 - h0) / 2., scale=std, size=num**2)
    xid = (xx // (Lx / num)).astype(int)
    yid = (yy // (Ly / num)).astype(int)
    bx, by = np.pi / (Lx / num), np.pi / (Ly / num)
    h = np.full_like(xx, h1)
    dh_dx = np.zeros_like(h)
    dh_dy = np.zeros_like(h)
    for k, hm in enumerate(hmins):
        m = (xid == k // num) & (yid == k % num)
        cx, cy = np.mean(xx[m]), np.mean(yy[m])
        h[m] -= (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
        dh_dx[m] += bx * (h1 - hm) * (np.sin(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
        dh_dy[m] += by * (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.sin(by * (yy[m] - cy)))
    return h, dh_dx, dh_dy


_PROFILES_1D = {'journal': journal_bearing, 'inclined': inclined_slider, 'parabolic': parabolic_slider, 'cdc': cdc}


def topography_rows(grid, geo, rows, hmins=None):
    """h, dh/dx, dh/dy on rows `rows` (indices into the (Nx+2)-row array incl. ghost rows) of the WHOLE domain's
    topography, shape (3, len(rows), Ny+2), without building the whole array: a slab of an 8192^2 problem needs 1026
    rows, not a 2 GB table per rank.  The 1-D profiles are functions of x alone; an asperity's centre is the mean of the
    cell centres it covers, which factorises into 1-D means (equal to the 2-D mean of `asperity` up to rounding).
    `hmins` (num^2 minimum heights) must be given for num > 1 -- they are random draws every rank has to share."""
    if geo.get('flip'):
        raise NotImplementedError("row-wise topography of a flipped geometry")
    rows = np.asarray(rows, int)
    Lx, Ly, Nx, Ny = grid['Lx'], grid['Ly'], grid['Nx'], grid['Ny']
    x_all = np.arange(-1, Nx + 1) / Nx * Lx + (Lx / Nx) / 2.
    y = np.arange(-1, Ny + 1) / Ny * Ly + (Ly / Ny) / 2.
    xx, yy = np.meshgrid(x_all[rows], y, indexing='ij')
    if geo['type'] in _PROFILES_1D:
        h, dh_dx, dh_dy = _PROFILES_1D[geo['type']](xx, grid, geo)
    elif geo['type'] == 'asperity':
        h0, h1, num = geo['hmin'], geo['hmax'], geo['num']
        if hmins is None:
            if num != 1:
                raise ValueError("topography_rows: pass the shared minimum heights of the num^2 asperities")
            hmins = np.array([h0])
        xid_all, yid_all = (x_all // (Lx / num)).astype(int), (y // (Ly / num)).astype(int)
        xid, yid = (xx // (Lx / num)).astype(int), (yy // (Ly / num)).astype(int)
        bx, by = np.pi / (Lx / num), np.pi / (Ly / num)
        h, dh_dx, dh_dy = np.full_like(xx, h1), np.zeros_like(xx), np.zeros_like(xx)
        for k, hm in enumerate(hmins):
            m = (xid == k // num) & (yid == k % num)
            if not m.any():
                continue
            cx, cy = np.mean(x_all[xid_all == k // num]), np.mean(y[yid_all == k % num])
            h[m] -= (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
            dh_dx[m] += bx * (h1 - hm) * (np.sin(bx * (xx[m] - cx)) * np.cos(by * (yy[m] - cy)))
            dh_dy[m] += by * (h1 - hm) * (np.cos(bx * (xx[m] - cx)) * np.sin(by * (yy[m] - cy)))
    else:
        raise IOError("Specify a valid geometry type")
    return np.stack([h, dh_dx, dh_dy])


class Topography:
    """Holds x, y, h, dh/dx, dh/dy (+ a zero deformation slot) as host arrays; the solver uploads
    ``full[:3]`` once (topography.py:180-255)."""

    def __init__(self, grid, geo, prop, on_change=None):
        xx, yy = create_midpoint_grid(grid)
        self._x, self._y = xx, yy
        self.dx, self.dy = grid['dx'], grid['dy']
        # with elastic deformation the device owns h, dh/dx, dh/dy and the displacement after the first update; the host
        # arrays are then a mirror that `refresh` (set by the Problem) brings up to date on access
        self.elastic = bool(prop.get('elastic', {}).get('enabled', False))
        self.refresh = None
        self._stale = False
        if geo['type'] in _PROFILES_1D:
            h, dh_dx, dh_dy = _PROFILES_1D[geo['type']](xx, grid, geo)
        elif geo['type'] == 'asperity':
            h, dh_dx, dh_dy = asperity(xx, yy, grid, geo)
        else:
            raise IOError("Specify a valid geometry type")
        self._field = np.zeros((4,) + xx.shape)
        ix, iy = (2, 1) if geo['flip'] else (1, 2)      # topography.py:227-234
        if geo['flip']:
            h, dh_dx, dh_dy = h.T, dh_dx.T, dh_dy.T
        self._field[0], self._field[ix], self._field[iy] = h, dh_dx, dh_dy
        self._on_change = on_change

    def update(self):
        """Topography.update of the reference (topography.py:257-271) happens on the device (gpf_elastic_update, called
        by Problem.update); nothing to do on the host."""

    def mark_stale(self):
        self._stale = True

    def _sync(self):
        if self._stale and self.refresh is not None:
            self._stale = False
            self.refresh(self._field)

    def update_gradients(self):
        # topography.py:273-280: second-order central differences, one-sided at the array edges
        self._field[1] = np.gradient(self._field[0], axis=0) / self.dx
        self._field[2] = np.gradient(self._field[0], axis=1) / self.dy
        if self._on_change:
            self._on_change()

    @property
    def full(self):
        self._sync()
        return self._field

    @property
    def h(self):
        self._sync()
        return self._field[0]

    @h.setter
    def h(self, value):
        # Assigning `topo.h = ...` re-differentiates and uploads (topography.py:273-280).  In-place edits of the arrays
        # returned by `h` / `full` stay on the host until `update_gradients()` (or a re-assignment) is called.
        self._sync()                # bring a stale mirror of an elastic gap up to date first, or it would overwrite `value`
        self._field[0] = value
        self.update_gradients()

    @property
    def deformation(self):
        self._sync()
        return self._field[3]

    @property
    def dh_dx(self):
        self._sync()
        return self._field[1]

    @property
    def dh_dy(self):
        self._sync()
        return self._field[2]

    @property
    def x(self):
        return self._x

    @property
    def y(self):
        return self._y
