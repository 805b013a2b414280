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
    """Converging - flat - diverging channel (topography.py:107-130): piecewise linear in x, flat lands of width b at both
    ends and of width 2 b in the middle."""
    half, land = grid['Lx'] / 2, geo['b']
    low, high = geo['hmin'], geo['hmax']
    slope = (high - low) / (half - 2 * land)
    zones = [(xx >= land) & (xx < half - land),             # converging
             (xx >= half - land) & (xx < half + land),      # flat centre
             (xx >= half + land) & (xx < grid['Lx'] - land)]  # diverging
    h = np.select(zones, [high - slope * (xx - land), low, low + slope * (xx - (half + land))], default=high)
    return h, np.select(zones, [-slope, 0., slope], default=0.), np.zeros_like(xx)


def _asperity_heights(geo):
    """Minimum gap under each of the num^2 asperities (topography.py:141-146): hmin for a single one, otherwise draws from
    an unseeded normal distribution around the mid-height, exactly as the reference does."""
    low, high, n = geo['hmin'], geo['hmax'], geo['num']
    if n == 1:
        return np.array([low])
    return np.random.normal(loc=low + (high - low) / 2., scale=(high - low) / 2. / 2.57, size=n**2)


def _asperity_field(xx, yy, grid, geo, heights, centre):
    """Cosine bumps on an n x n checkerboard (topography.py:148-170).  `centre(k, cells)` returns the centre of bump k, whose
    cells are selected by the boolean array `cells`; bump k sits in checkerboard column k // n, row k % n."""
    n, high = geo['num'], geo['hmax']
    wx, wy = np.pi / (grid['Lx'] / n), np.pi / (grid['Ly'] / n)
    col, row = (xx // (grid['Lx'] / n)).astype(int), (yy // (grid['Ly'] / n)).astype(int)
    h, gx, gy = np.full_like(xx, high), np.zeros_like(xx), np.zeros_like(xx)
    for k, low in enumerate(heights):
        cells = (col == k // n) & (row == k % n)
        if not cells.any():
            continue
        cx, cy = centre(k, cells)
        depth, u, v = high - low, wx * (xx[cells] - cx), wy * (yy[cells] - cy)
        h[cells] -= depth * (np.cos(u) * np.cos(v))
        gx[cells] += wx * depth * (np.sin(u) * np.cos(v))
        gy[cells] += wy * depth * (np.cos(u) * np.sin(v))
    return h, gx, gy


def asperity(xx, yy, grid, geo):
    # a bump's centre is the mean of the cell centres it covers (topography.py:158-159)
    return _asperity_field(xx, yy, grid, geo, _asperity_heights(geo), lambda k, cells: (np.mean(xx[cells]), np.mean(yy[cells])))


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
        n = geo['num']
        if hmins is None:
            if n != 1:
                raise ValueError("topography_rows: pass the shared minimum heights of the num^2 asperities")
            hmins = np.array([geo['hmin']])
        # the centre of a bump from ALL rows / columns it covers, not only from the requested rows
        col_all, row_all = (x_all // (Lx / n)).astype(int), (y // (Ly / n)).astype(int)
        h, dh_dx, dh_dy = _asperity_field(xx, yy, grid, geo, hmins,
                                          lambda k, cells: (np.mean(x_all[col_all == k // n]), np.mean(y[row_all == k % n])))
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
