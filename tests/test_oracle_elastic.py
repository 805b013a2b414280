"""Physics pins of the elastic half-space restatement (oracle/elastic.py; the reference's own arithmetic lives in
ContactMechanics, absent here -> PARITY UNPINNED against it)."""
import numpy as np
import pytest

from oracle.elastic import ElasticDeformation, love_kernel


def _grid(Nx, Ny, Lx, Ly, perX, perY):
    g = {'Nx': Nx, 'Ny': Ny, 'Lx': Lx, 'Ly': Ly, 'dx': Lx / Nx, 'dy': Ly / Ny}
    for side, per in (('xE', perX), ('xW', perX), ('yS', perY), ('yN', perY)):
        g[f'bc_{side}_P'] = [per] * 3
    return g


def test_periodic_sinusoidal_load_gives_the_textbook_amplitude():
    """u = 2 p0 cos(qx) / (E* q) for p = p0 cos(qx) (Johnson, Contact Mechanics, section 13.2), times the reference's
    cell-area ratio (its grid of Nx+2 points spans Lx)."""
    g = _grid(62, 30, 1e-3, 5e-4, True, True)
    E, v = 210e9, 0.3
    el = ElasticDeformation(E, v, 1.0, g, 10)
    nx, ny = 64, 32
    x = np.arange(nx) * (g['Lx'] / nx)
    k = 3
    p = 1e6 * np.cos(2 * np.pi * k * x / g['Lx'])[:, None] * np.ones((1, ny))
    u = el.get_deformation(p)
    q = 2 * np.pi * k / g['Lx']
    expect = 2 * p / (E / (1 - v**2) * q) * (g['dx'] * g['dy'] / el.area_per_pt)
    np.testing.assert_allclose(u, expect, rtol=1e-12, atol=1e-22)
    assert el.periodicity == 'full' and abs(u.mean()) < 1e-20          # q = 0 mode removed


@pytest.mark.parametrize('perX,perY', [(False, False), (False, True), (True, False)])
def test_aperiodic_convolution_equals_direct_summation(perX, perY):
    g = _grid(10, 6, 2e-4, 1.2e-4, perX, perY)
    el = ElasticDeformation(50e9, 0.3, 1.0, g, 2)
    nx, ny = 12, 8
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 1e6, (nx, ny))
    u = el.get_deformation(p)
    sx, sy = g['Lx'] / nx, g['Ly'] / ny
    young = 50e9 / (1 - 0.09)
    direct = np.zeros((nx, ny))
    for i in range(nx):
        for j in range(ny):
            for a in range(nx):
                for b in range(ny):
                    dxs, dys = (i - a) * sx, (j - b) * sy
                    images_x = [k * g['Lx'] for k in range(-2, 3)] if perX else [0.0]
                    images_y = [k * g['Ly'] for k in range(-2, 3)] if perY else [0.0]
                    if perX:
                        dxs = ((i - a + nx // 2) % nx - nx // 2) * sx if (i - a) % nx <= nx // 2 else ((i - a) % nx - nx) * sx
                    if perY:
                        dys = ((j - b + ny // 2) % ny - ny // 2) * sy if (j - b) % ny <= ny // 2 else ((j - b) % ny - ny) * sy
                    s = 0.0
                    for ox in images_x:
                        for oy in images_y:
                            s += love_kernel(np.float64(dxs + ox), np.float64(dys + oy), sx / 2, sy / 2, young)
                    direct[i, j] += s * p[a, b]
    direct *= g['dx'] * g['dy'] / (sx * sy)
    np.testing.assert_allclose(u, direct, rtol=1e-9)
    assert el.periodicity == ('none' if not (perX or perY) else 'half')


def test_love_kernel_centre_value_and_far_field():
    """Centre of a uniformly loaded square: u = 4 a ln(1 + sqrt 2) * 2 / (pi E*) ... (Johnson eq. 3.27); far away the patch
    acts like a point force P / (pi E* r) (Boussinesq)."""
    a, young = 1e-5, 1e11
    centre = love_kernel(np.float64(0.), np.float64(0.), a, a, young)
    np.testing.assert_allclose(centre, 8 * a * np.log(1 + np.sqrt(2)) / (np.pi * young), rtol=1e-12)
    r = 400 * a
    np.testing.assert_allclose(love_kernel(np.float64(r), np.float64(0.), a, a, young), 4 * a * a / (np.pi * young * r), rtol=1e-5)


def test_underrelaxation_and_reference_point():
    g = _grid(10, 6, 2e-4, 1.2e-4, False, False)
    el = ElasticDeformation(50e9, 0.3, 0.25, g, 0)
    p = np.random.default_rng(1).uniform(0, 1e6, (12, 8))
    full = ElasticDeformation(50e9, 0.3, 1.0, g, 0).get_deformation(p - p[0, 0])
    d1 = el.update(p)
    np.testing.assert_allclose(d1, 0.25 * (full - full[0, 0]), rtol=1e-12, atol=1e-24)
    d2 = el.update(p)
    np.testing.assert_allclose(d2, (0.75 * 0.25 + 0.25) * (full - full[0, 0]), rtol=1e-12, atol=1e-24)
    assert d2[0, 0] == 0.0
