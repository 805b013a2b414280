"""Problem-class-agnostic restatement of the reference's solver tests (see test_oracle_reference_suite.py).

`make(d)` builds a problem (oracle or GPU) from a sanitised input dict, `reader` is the matching
read_yaml_input.  Inputs are the reference tests' YAML strings; assertions and tolerances are theirs.
"""
import io
from copy import deepcopy

import numpy as np

JOURNAL_1D = """
options:
    output: data/journal
    write_freq: 1000
    silent: True
grid:
    dx: 1.e-5
    dy: 1.
    Nx: 100
    Ny: 1
    xE: ['P', 'P', 'P']
    xW: ['P', 'P', 'P']
    yS: ['P', 'P', 'P']
    yN: ['P', 'P', 'P']
geometry:
    type: journal
    CR: 1.e-2
    eps: 0.7
    U: 0.1
    V: 0.
numerics:
    CFL: 0.5
    adaptive: 1
    tol: 1e-8
    dt: 1e-10
    max_it: 10_000
properties:
    shear: 0.0794
    bulk: 0.
    EOS: DH
    P0: 101325.
    rho0: 877.7007
    T0: 323.15
    C1: 3.5e12
    C2: 1.23
"""

JOURNAL_2D = JOURNAL_1D.replace("dy: 1.\n", "dy: {dy}\n").replace("dx: 1.e-5", "dx: {dx}") \
    .replace("Nx: 100", "Nx: {n}").replace("Ny: 1\n", "Ny: {n}\n").replace("C1: 3.5e12", "C1: 3.5e10")

DECAY = """
options:
    output: data/decay
    write_freq: 100
    use_tstamp: False
    silent: True
grid:
    Lx: 3.2e-7
    Ly: 1
    Nx: 256
    Ny: 1
    xE: ['P', 'P', 'P']
    xW: ['P', 'P', 'P']
    yS: ['P', 'P', 'P']
    yN: ['P', 'P', 'P']
geometry:
    type: inclined
    hmin: 5e-9
    hmax: 5e-9
    U: 0.
    V: 0.
numerics:
    adaptive: 0
    CFL: 0.5
    dt: 1e-13
    max_it: 5_000
properties:
    EOS: cubic
    shear: 3.92293e-05
    bulk: 0.
    rho0: 762.8617
    a: 1.33030e-1
    b: -1.41778e2
    c: 8.35134e4
    d: -2.86532e6
"""


def _read(reader, text):
    with io.StringIO(text) as f:
        return reader(f)


def sommerfeld_solution(x, Lx, mu, U, clearance_ratio, eps, P0):
    # tests/test_sommerfeld.py:70-104
    Rb = Lx / (2. * np.pi)
    c = clearance_ratio * Rb
    omega = U / Rb
    prefac = 6. * mu * omega * (Rb / c)**2 * eps
    return P0 + prefac * np.sin(x / Rb) * (2. + eps * np.cos(x / Rb)) / ((2. + eps**2) * (1. + eps * np.cos(x / Rb))**2)


def _pressure(problem):
    p = problem.pressure
    return p.pressure if hasattr(p, 'pressure') else p


def check_sommerfeld(make, reader, eps):
    d = _read(reader, JOURNAL_1D)
    d['geometry']['eps'] = eps
    problem = make(d)
    problem.run()
    if hasattr(problem, 'update_closures'):
        problem.update_closures()
    p_num = _pressure(problem)[1:-1, 1]
    Lx = problem.grid['Lx']
    x_ana = np.linspace(0., Lx, 101)
    x_num = (x_ana[1:] + x_ana[:-1]) / 2.
    dp = p_num[1] - p_num[0]
    p_ana = sommerfeld_solution(x_num, Lx, problem.prop['shear'], problem.geo['U'], problem.geo['CR'], eps, p_num[0] - dp / 2)
    rel_err = np.linalg.norm(p_ana - p_num) / np.linalg.norm(p_ana)
    assert rel_err < 0.02
    return problem


def _x(problem):
    return (problem.topo.x if hasattr(problem.topo, 'x') else problem.x)[1:-1, 1]


def check_shear_wave_decay(make, reader, n):
    problem = make(_read(reader, DECAY))
    problem._pre_run()
    h = problem.geo['hmin']
    kin_visc = problem.prop['shear'] / problem.prop['rho0']
    kn = n * 2. * np.pi / problem.grid['Lx']
    tau = h**2 / (6 * kin_visc)
    x = _x(problem)
    problem.q[2, 1:-1, :] = np.sin(kn * x)[:, None]
    problem.kinetic_energy_old = problem.kinetic_energy
    for _ in range(200):
        problem.update()
        jy_ana = np.sin(kn * x) * np.exp(-2 * problem.simtime / tau)
        np.testing.assert_almost_equal(problem.q[2, 1:-1, 1], jy_ana, decimal=4)


def check_sound_wave_decay(make, reader, n):
    problem = make(_read(reader, DECAY))
    problem._pre_run()
    h = problem.geo['hmin']
    kin_visc = problem.prop['shear'] / problem.prop['rho0']
    kn = n * 2. * np.pi / problem.grid['Lx']
    tau = h**2 / (6 * kin_visc)
    cT = problem.pressure.v_sound if hasattr(problem.pressure, 'v_sound') else problem.v_sound
    x = _x(problem)
    problem.q[1, 1:-1, :] = np.sin(kn * x)[:, None]
    problem.kinetic_energy_old = problem.kinetic_energy
    k_crit = 6. * kin_visc / (h**2 * cT)
    for _ in range(400):
        problem.update()
        t = problem.simtime
        if kn > k_crit:
            sT = np.sqrt(cT**2 - (1 / tau / kn)**2)
            amp = np.exp(-t / tau) * (np.cos(sT * kn * t) - 1 / (tau * sT * kn) * np.sin(sT * kn * t))
        else:
            isT = np.sqrt((1 / tau / kn)**2 - cT**2)
            amp = np.exp(-t / tau) * (np.cosh(isT * kn * t) - 1 / (tau * isT * kn) * np.sinh(isT * kn * t))
        np.testing.assert_almost_equal(problem.q[1, 1:-1, 1], np.sin(kn * x) * amp, decimal=3)


def check_mass_conservation(make, reader):
    problem = make(_read(reader, JOURNAL_2D.format(dx='2.e-5', dy='2.e-5', n=50)))
    problem._pre_run()
    mass_before = float(problem.mass)
    for _ in range(50):
        problem.update()
    assert np.isclose(problem.mass, mass_before)


def check_flip_axes(make, reader, n=100):
    input_x = _read(reader, JOURNAL_2D.format(dx='1.e-5', dy='1.e-5', n=n))
    input_y = deepcopy(input_x)
    input_y['geometry']['U'] = 0.
    input_y['geometry']['V'] = input_x['geometry']['U']
    input_y['geometry']['flip'] = True
    px, py = make(input_x), make(input_y)
    px._pre_run()
    py._pre_run()
    for _ in range(5):
        px.update()
        py.update()
        np.testing.assert_almost_equal(px.q[0, 1:-1, 1:-1], py.q[0, 1:-1, 1:-1].T)
        np.testing.assert_almost_equal(px.q[1, 1:-1, 1:-1], py.q[2, 1:-1, 1:-1].T)
        np.testing.assert_almost_equal(px.q[2, 1:-1, 1:-1], py.q[1, 1:-1, 1:-1].T)
