"""Parity at BASELINE.json's full grid sizes through size-independent properties (the oracle would need
~40 s per step at 4096^2): conservation, symmetry, dimensional reduction, batching invariance."""
import io
from copy import deepcopy

import numpy as np
import pytest

import reference_suite as rs

pytestmark = pytest.mark.gpu


def make(text_or_dict):
    from gapflow_amd import Problem
    from gapflow_amd.io import read_yaml_input
    d = read_yaml_input(io.StringIO(text_or_dict)) if isinstance(text_or_dict, str) else text_or_dict
    return Problem._from_dict(d), d


def test_journal_4096_mass_yinvariance_and_1d_reduction(hiplib):
    """configs[2] (4096^2 periodic journal): mass is conserved, the field stays exactly y-invariant (the geometry is),
    and every column equals the 1-D problem (4096 x 1) with the same dy advanced by the same kernels."""
    text = rs.JOURNAL_2D.format(dx='1.e-5', dy='1.e-5', n=4096)
    p2, _ = make(text)
    p1, _ = make(rs.JOURNAL_1D.replace('Nx: 100', 'Nx: 4096').replace('dy: 1.', 'dy: 1.e-5').replace('C1: 3.5e12', 'C1: 3.5e10'))
    p2._pre_run()
    p1._pre_run()
    m0 = p2.mass
    p2._advance(20, honor_stop=False)
    p1._advance(20, honor_stop=False)
    assert p2.step == 20 and p1.step == 20
    np.testing.assert_allclose(p2.mass, m0, rtol=1e-12)
    q = p2.q
    assert np.isfinite(q).all()
    assert np.array_equal(q[:, :, 1:], q[:, :, :-1]), 'field lost its y-invariance'
    np.testing.assert_array_equal(q[2], 0.0)
    np.testing.assert_allclose(p2.dt, p1.dt, rtol=1e-13)
    for c in range(2):
        np.testing.assert_allclose(q[c, :, 1], p1.q[c, :, 1], rtol=1e-12, atol=0)
    # Ekin sums 4098 copies of the 1-D problem's 3 columns
    np.testing.assert_allclose(p2.kinetic_energy / 4098., p1.kinetic_energy / 3., rtol=1e-12)


def test_flip_symmetry_2048(hiplib):
    """tests/test_flip_axes.py at 2048^2: the y-driven transposed problem is the mirror image of the x-driven one."""
    _, dx = make(rs.JOURNAL_2D.format(dx='1.e-5', dy='1.e-5', n=2048))
    dy = deepcopy(dx)
    dy['geometry']['U'], dy['geometry']['V'], dy['geometry']['flip'] = 0., dx['geometry']['U'], True
    px, _ = make(dx)
    py, _ = make(dy)
    px._pre_run()
    py._pre_run()
    px._advance(5, honor_stop=False)
    py._advance(5, honor_stop=False)
    for a, b in ((0, 0), (1, 2), (2, 1)):
        x, y = px.q[a, 1:-1, 1:-1], py.q[b, 1:-1, 1:-1].T
        scale = np.abs(x).max() or 1.
        assert np.abs(x - y).max() <= 1e-9 * scale + 1e-7


def test_slider_1024_batching_invariance_and_ghost_rules(hiplib):
    """configs[1] (1024^2 inclined slider, D/N in x): 40 steps in one batch == 4 batches of 10, and the ghost cells of
    the result obey the boundary rules."""
    text = """
options: {silent: True}
grid: {Nx: 1024, Ny: 1024, Lx: 0.1, Ly: 0.1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 1.e-5, U: 50., V: 0.}
numerics: {CFL: 0.4, adaptive: 1, tol: 1.e-12, max_it: 100000}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007}
"""
    a, _ = make(text)
    b, _ = make(text)
    a._pre_run()
    b._pre_run()
    a._advance(40, honor_stop=False)
    for _ in range(4):
        b._advance(10, honor_stop=False)
    assert np.array_equal(a.q, b.q) and a.dt == b.dt and a.simtime == b.simtime
    assert np.isfinite(a.q).all() and (a.q[0] > 0).all()
    # density ghost cells obey the Dirichlet rule, flux ghost cells the Neumann rule (problem.py:758-766)
    q = a.q
    np.testing.assert_allclose(q[0, 0, 1:-1] + q[0, 1, 1:-1], 2 * 877.7007, rtol=1e-14)
    np.testing.assert_array_equal(q[1, -1, 1:-1], q[1, -2, 1:-1])
    np.testing.assert_array_equal(q[:, :, 0], q[:, :, -2])
