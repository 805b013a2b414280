"""The inputs the reference ships as examples (examples/config/*.yaml, examples/slip_1d_lj_mock.py), run end to end
through the drop-in API on the GPU.  The parameter sets are restated here as data (the reference tree does not travel
to the GPU box); `max_it` is cut so that each case takes seconds.  What is asserted is what a user switching over
needs: the input is accepted with the reference's defaults, run() completes, fields stay finite, the output files
and the history have the reference's shape, and the surrogate cases train / extend their database.

LAMMPS-driven examples (journal_1d_gold-hexadecane_gp_lammps, parabolic_1d_lj_gp_lammps) are out of scope (DESIGN.md
section 7); the elastic one (parabolic_1d_elastic) runs in tests/test_gpu_elastic.py."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DN = "xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']"

INCLINED_PL = """
options: {{output: {out}, write_freq: 1000, silent: False}}
grid: {{Lx: 0.1, Ly: 1., Nx: 100, Ny: 1, %s, xE_D: 1.1853, xW_D: 1.1853}}
geometry: {{type: inclined, hmax: 6.6e-5, hmin: 1e-5, U: 50., V: 0.}}
numerics: {{CFL: 0.4, adaptive: True, tol: 1e-6, dt: 1e-8, max_it: 3000}}
properties: {{EOS: PL, shear: 1.846e-5, bulk: 0., P0: 101325, rho0: 1.1853, alpha: 0.}}
""" % DN

JOURNAL_DH = """
options: {{output: {out}, write_freq: 1000, use_tstamp: True}}
grid: {{dx: 1.e-5, dy: 1., Nx: 100, Ny: 1, xE: ['P', 'P', 'P'], xW: ['P', 'P', 'P'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{CFL: 0.25, adaptive: 1, tol: 1e-9, dt: 1e-10, max_it: 2000}}
properties: {{shear: 0.0794, bulk: 0., EOS: DH, P0: 101325, rho0: 877.7007, T0: 323.15, C1: 3.5e10, C2: 1.23}}
"""

PARABOLIC_BAYADA = """
options: {{output: {out}, write_freq: 1000, use_tstamp: True}}
grid: {{Lx: 0.0762, Ly: 1., Nx: 100, Ny: 1, %s, xE_D: 850., xW_D: 850.}}
geometry: {{type: parabolic, hmin: 2.54e-5, hmax: 5.08e-5, U: 4.57, V: 0.}}
numerics: {{adaptive: 1, CFL: 0.45, tol: 1e-7, dt: 1.e-10, max_it: 3000}}
properties:
    EOS: Bayada
    rho0: 850.
    shear: 0.039
    bulk: 0.
    cl: 1600.
    cv: 352.
    piezo: {{name: Dukler, shearv: 3.9e-5, rhol: 850., rhov: 0.019}}
""" % DN

GP_BLOCK = """
gp:
    press: {{fix_noise: True, atol: {pa}, rtol: {pr}, obs_stddev: {ps}, max_steps: {ms}}}
    shear: {{fix_noise: True, atol: {pa}, rtol: {pr}, obs_stddev: {ss}, max_steps: {ms}}}
db: {{dtool: True, init_size: 5, init_method: {im}, init_width: {iw}}}
"""

JOURNAL_1D_GP = """
options: {{output: {out}, write_freq: 100, use_tstamp: True}}
grid: {{dx: 1.e-5, dy: 1., Nx: 100, Ny: 1, %s, xE_D: 877.7007, xW_D: 877.7007}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{CFL: 0.25, adaptive: 1, tol: 1e-9, dt: 1e-10, max_it: 150}}
properties: {{shear: 0.0794, bulk: 0., EOS: DH, P0: 101325, rho0: 877.7007, T0: 323.15, C1: 3.5e10, C2: 1.23}}
""" % DN + GP_BLOCK.format(pa=1., pr=0.1, ps=100., ss=1., ms=5, im='lhc', iw='1.e-6').replace('{', '{{').replace('}', '}}')

JOURNAL_2D_GP = """
options: {{output: {out}, write_freq: 100, use_tstamp: True}}
grid: {{dx: 1.e-5, dy: 1.e-5, Nx: 100, Ny: 100, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'],
       xE_D: 877.7007, xW_D: 877.7007, yN_D: 877.7007, yS_D: 877.7007}}
geometry: {{type: journal, CR: 1.e-2, eps: 0.7, U: 0.1, V: 0.}}
numerics: {{CFL: 0.25, adaptive: 1, tol: 1e-9, dt: 1e-10, max_it: 120}}
properties: {{shear: 0.0794, bulk: 0., EOS: DH, P0: 101325, rho0: 877.7007, T0: 323.15, C1: 3.5e10, C2: 1.23}}
""" + GP_BLOCK.format(pa=1., pr=0.1, ps=100., ss=1., ms=5, im='rand', iw='1.e-6').replace('{', '{{').replace('}', '}}')

LJ_PROPS = "properties: {{shear: 2.15, bulk: 0., EOS: BWR, T: 1.0, rho0: 0.8}}"

PARABOLIC_LJ_GP = """
options: {{output: {out}, write_freq: 100, use_tstamp: True}}
grid: {{Lx: 1470., Ly: 1., Nx: 200, Ny: 1, %s, xE_D: 0.8, xW_D: 0.8}}
geometry: {{type: parabolic, hmin: 12., hmax: 60., U: 0.12, V: 0.}}
numerics: {{CFL: 0.5, adaptive: 1, tol: 1e-8, dt: 0.05, max_it: 150}}
%s
gp:
    press: {{fix_noise: True, atol: 1.5, rtol: 0., obs_stddev: 2.e-2, max_steps: 10, active_learning: True}}
    shear: {{fix_noise: True, atol: 1.5, rtol: 0., obs_stddev: 4.e-3, max_steps: 10, active_learning: True}}
db: {{init_size: 5, init_method: rand, init_width: 0.01}}
""" % (DN, LJ_PROPS)

ASPERITY_LJ_GP = """
options: {{output: {out}, write_freq: 100, use_tstamp: True}}
grid: {{Lx: 1470., Ly: 1470., Nx: 100, Ny: 100, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], yS: ['D', 'N', 'N'], yN: ['D', 'N', 'N'],
       xE_D: 0.8, xW_D: 0.8, yS_D: 0.8, yN_D: 0.8}}
geometry: {{type: asperity, hmin: 12., hmax: 60., U: 0.12, V: 0.}}
numerics: {{CFL: 0.5, adaptive: 1, tol: 1e-8, dt: 0.05, max_it: 100}}
%s
gp:
    press: {{fix_noise: True, atol: 1.5, rtol: 0., obs_stddev: 2.e-2, max_steps: 10}}
    shear: {{fix_noise: True, atol: 1.5, rtol: 0., obs_stddev: 4.e-3, max_steps: 10}}
db: {{init_size: 5, init_method: rand, init_width: 0.01}}
""" % LJ_PROPS


def _outdir(tmp_path):
    runs = [d for d in os.listdir(tmp_path) if os.path.isdir(tmp_path / d)]
    assert len(runs) == 1, runs
    return tmp_path / runs[0]


@pytest.mark.parametrize('name,text', [('inclined_1d_powerlaw', INCLINED_PL), ('journal_1d_dowson-higginson', JOURNAL_DH),
                                       ('parabolic_1d_cav_bayada', PARABOLIC_BAYADA)],
                         ids=['inclined_1d_powerlaw', 'journal_1d_dowson-higginson', 'parabolic_1d_cav_bayada'])
def test_fixed_form_examples_run(hiplib, tmp_path, name, text):
    from gapflow_amd import Problem
    prob = Problem.from_string(text.format(out=str(tmp_path / name)))
    prob.run()
    assert prob.step == prob.numerics['max_it'] or prob.converged
    assert np.isfinite(prob.q).all() and np.isfinite(prob.pressure.pressure).all()
    assert prob.q[0].min() > 0
    out = _outdir(tmp_path)
    for f in ('config.yml', 'history.csv', 'sol.nc', 'topo.nc'):
        assert (out / f).exists(), f
    rows = open(out / 'history.csv').read().strip().splitlines()
    assert rows[0].split(',') == ['step', 'time', 'ekin', 'residual', 'vsound']
    assert len(rows) >= 2
    # a lubricated contact builds up pressure above ambient somewhere in the converging part of the gap
    if name != 'journal_1d_dowson-higginson':
        assert prob.pressure.pressure.max() > prob.prop.get('P0', 0.)


@pytest.mark.parametrize('name,text,dim', [('journal_1d_gp', JOURNAL_1D_GP, 1), ('journal_2d_gp', JOURNAL_2D_GP, 2),
                                           ('parabolic_1d_lj_gp', PARABOLIC_LJ_GP, 1), ('asperity_2d_lj_gp', ASPERITY_LJ_GP, 2)],
                         ids=['journal_1d_gp', 'journal_2d_gp', 'parabolic_1d_lj_gp', 'asperity_2d_lj_gp'])
def test_surrogate_examples_run(hiplib, tmp_path, name, text, dim):
    """`db:` without `md:` attaches the Mock runner (problem.py:232-243); active learning defaults to True (io.py:416)."""
    from gapflow_amd import Problem
    prob = Problem.from_string(text.format(out=str(tmp_path / name)))
    assert set(prob._gp_models) == ({'zz', 'xz'} if dim == 1 else {'zz', 'xz', 'yz'})
    prob.run()
    assert prob.step == prob.numerics['max_it'] or prob.converged or prob._stop
    assert np.isfinite(prob.q).all()
    assert prob.database.size >= 5
    for m in prob._gp_models.values():
        assert m.last_fit_train_size == prob.database.size           # every model is trained on the final database
        assert np.all(np.isfinite(m.kernel_lengthscale)) and m.kernel_variance > 0
    out = _outdir(tmp_path)
    for f in ('config.yml', 'history.csv', 'sol.nc', 'topo.nc'):
        assert (out / f).exists(), f
    assert any(f.startswith('gp_') and f.endswith('.csv') for f in os.listdir(out)), os.listdir(out)


def test_slip_example_with_extra_field(hiplib, tmp_path):
    """examples/slip_1d_lj_mock.py: explicit Mock + Database + Problem(..., extra_field=...) with the slip length as a
    GP input dimension (active_dims x: [0, 1, 6]) and a density-only pressure model (active_dims [0])."""
    from scipy.special import erf
    from gapflow_amd.problem import Problem
    from gapflow_amd.io import read_yaml_input
    from gapflow_amd.db import Database
    from gapflow_amd.md import Mock
    text = """
options: {output: %s, write_freq: 100, use_tstamp: False}
grid: {Lx: 1470., Ly: 1., Nx: 200, Ny: 1, xE: ['P', 'P', 'P'], xW: ['P', 'P', 'P'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: inclined, hmin: 12., hmax: 12., U: 0.12, V: 0.}
numerics: {CFL: 0.5, adaptive: 1, tol: 1e-8, dt: 0.1, max_it: 120}
properties: {shear: 2.15, bulk: 0., EOS: BWR, T: 1.0, rho0: 0.8}
gp:
    press: {fix_noise: True, atol: 1., rtol: 0., obs_stddev: 2.e-2, max_steps: 10, active_dims: [0, ]}
    shear:
        fix_noise: True
        atol: 1.
        rtol: 0.
        obs_stddev: 4.e-3
        max_steps: 10
        active_dims: {x: [0, 1, 6]}
db: {init_size: 10, init_method: lhc}
""" % str(tmp_path / 'slip_1d_lj')
    with io.StringIO(text) as f:
        d = read_yaml_input(f)
    nx, ny, a = d['grid']['Nx'], d['grid']['Ny'], 20.
    slip = np.zeros(nx)
    e = erf(np.linspace(-a, a, nx // 2))
    slip[:nx // 2], slip[nx // 2:] = e, -e
    slip = (1. + np.roll(slip, nx // 4)) / 2.
    extra = np.zeros((1, nx + 2, ny + 2))
    extra[0, 1:-1, :] = slip[:, None]
    extra[0, 0, :], extra[0, -1, :] = extra[0, -2, :], extra[0, 1, :]
    database = Database(Mock(d['properties'], d['geometry'], d['gp']), d['db'])
    prob = Problem(d['options'], d['grid'], d['numerics'], d['properties'], d['geometry'], d['gp'], database,
                   extra_field=extra)
    prob.run()
    assert prob._gp_models['zz'].active_dims == [0] and prob._gp_models['xz'].active_dims == [0, 1, 6]
    assert np.isfinite(prob.q).all() and database.size >= 10
    # the slip length modulates the wall shear stress along x: the surrogate has to see it
    tau = prob.wall_stress_xz.lower[4, 1:-1, 1]
    assert np.ptp(tau) > 0
