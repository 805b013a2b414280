"""`gapflow_amd.models.{viscous,pressure,sound}` -- the reference's leaf functions as device operators -- against the
golden vectors (true outputs of the reference's leaf modules) through the C ABI (gpf_viscous_stress, gpf_eos)."""
import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

LEAF = np.load(GOLDEN + '/leaf_closures.npz')
SLIP = np.load(GOLDEN + '/leaf_viscous_slip.npz')
EOS_PROPS = {
    'DH': dict(EOS='DH', rho0=877.7007, P0=101325., C1=3.5e10, C2=1.23),
    'PL': dict(EOS='PL', rho0=1.1853, P0=101325., alpha=0.),
    'vdW': dict(EOS='vdW', M=39.948, T=100., a=1.355, b=0.03201),
    'MT': dict(EOS='MT', rho0=700., P0=0.101e6, K=0.557e9, n=7.33),
    'cubic': dict(EOS='cubic', a=1.33030e-1, b=-1.41778e2, c=8.35134e4, d=-2.86532e6),
    'BWR': dict(EOS='BWR', T=1.0, gamma=3.0),
    'Bayada': dict(EOS='Bayada', rho_l=850., rho_v=0.019, c_l=1600., c_v=352.),
}


def _err(got, ref):
    scale = np.maximum(np.abs(ref), np.abs(ref).max(axis=(1, 2), keepdims=True) * 1e-3) + 1e-30
    return float(np.max(np.abs(got - ref) / scale))


@pytest.mark.parametrize('slip', ['top', 'both', 'bottom', 'none'])
@pytest.mark.parametrize('grad', ['g0', 'g1'])
def test_viscous_operators_every_slip_keyword(hiplib, slip, grad):
    from gapflow_amd.models import viscous
    q, h, Ls = SLIP['q'], SLIP['h'], SLIP['Ls']
    U, V, eta, zeta = SLIP['params']
    gx, gy = (SLIP['dqx'], SLIP['dqy']) if grad == 'g1' else (None, None)
    for name in ('stress_bottom', 'stress_top', 'stress_avg'):
        got = getattr(viscous, name)(q, h, U, V, eta, zeta, Ls, dqx=gx, dqy=gy, slip=slip)
        ref = SLIP[f'{name}_{slip}_{grad}']
        assert got.shape == ref.shape
        assert _err(got, ref) <= 1e-9, name          # tolerance of the north star for fp64 fields


@pytest.mark.parametrize('tag', ['Ls0', 'LsF'])
def test_viscous_operators_solver_branch(hiplib, tag):
    """Same inputs as the oracle's solver-branch fixture (slip="top", no gradients, per-cell slip length as (1, nx, ny))."""
    from gapflow_amd.models import viscous
    q, h, Ls = LEAF['visc_q'], LEAF['visc_h'], LEAF[f'visc_{tag}_Ls']
    U, V, eta, zeta = LEAF['visc_params']
    for fn, key in ((viscous.stress_bottom, 'bot'), (viscous.stress_top, 'top'), (viscous.stress_avg, 'avg')):
        assert _err(fn(q, h, U, V, eta, zeta, Ls[0]), LEAF[f'visc_{tag}_{key}']) <= 1e-9, key


def test_viscous_operators_single_point_and_viscosity_field(hiplib):
    """tests/test_analytic.py:58-60 calls these with shape-(3,) points; stress.py:306-326 passes a viscosity field."""
    from gapflow_amd.models import viscous
    from oracle import closures as ocl
    q, h = np.array([1.0, 0.75, 0.25]), np.array([1.0, 0.01, 0.01])
    for slip in ('top', 'both'):
        for fn, ofn in ((viscous.stress_avg, ocl.stress_avg), (viscous.stress_top, ocl.stress_top), (viscous.stress_bottom, ocl.stress_bottom)):
            got = fn(q, h, U=1., V=1., eta=1., zeta=1., Ls=0.5, slip=slip)
            ref = ofn(q, h, 1., 1., 1., 1., 0.5, slip=slip)
            assert got.shape == ref.shape
            np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-14)
    qf, hf = SLIP['q'], SLIP['h']
    eta_field = 0.05 + 0.03 * np.random.default_rng(2).random(qf.shape[1:])
    got = viscous.stress_top(qf, hf, 0.1, -0.07, eta_field, 0.013, SLIP['Ls'], slip='both')
    ref = ocl.stress_top(qf, hf, 0.1, -0.07, eta_field, 0.013, SLIP['Ls'], slip='both')
    assert _err(got, ref) <= 1e-10


@pytest.mark.parametrize('eos', sorted(EOS_PROPS))
def test_eos_operators(hiplib, eos):
    from gapflow_amd.models.pressure import eos_pressure
    from gapflow_amd.models.sound import eos_sound_velocity
    rho = LEAF[f'eos_{eos}_rho']
    np.testing.assert_allclose(eos_pressure(rho, EOS_PROPS[eos]), LEAF[f'eos_{eos}_p'], rtol=1e-9)
    np.testing.assert_allclose(eos_sound_velocity(rho, EOS_PROPS[eos]), LEAF[f'eos_{eos}_c'], rtol=1e-9, equal_nan=True)


def test_eos_function_defaults(hiplib):
    """eos_pressure passes only the keys present in `prop`; the rest are the functions' defaults (pressure.py:73-76, 112)."""
    from gapflow_amd.models.pressure import eos_pressure
    rho = np.linspace(0.5, 2.0, 9)
    np.testing.assert_allclose(eos_pressure(rho, {'EOS': 'PL'}), 101325. * (rho / 1.1853), rtol=1e-12)
    with pytest.raises(TypeError):
        eos_pressure(rho, {'EOS': 'BWR'})            # bwr(dens, T, ...) has no default temperature


def test_viscosity_operators(hiplib):
    """models/viscosity.py against the golden vectors of the reference's leaf module."""
    from gapflow_amd.models import viscosity
    pz = {'Barus': dict(name='Barus', aB=20e-9), 'Roelands': dict(name='Roelands', mu_inf=1e-3, p_ref=1.96e8, z=0.68),
          'Dukler': dict(name='Dukler', eta_v=3.9e-5, rho_l=850., rho_v=0.019),
          'McAdams': dict(name='McAdams', eta_v=3.9e-5, rho_l=850., rho_v=0.019)}
    for k, d in pz.items():
        arg = LEAF['piezo_rho'] if k in ('Dukler', 'McAdams') else LEAF['piezo_p']
        np.testing.assert_allclose(viscosity.piezoviscosity(arg, 0.0794, d), LEAF[f'piezo_{k}'], rtol=1e-9)
    th = {'Eyring': dict(name='Eyring', tauE=5e5), 'Carreau': dict(name='Carreau', mu_inf=1e-9, lam=1e-6, a=2., N=0.6)}
    for k, d in th.items():
        np.testing.assert_allclose(viscosity.shear_thinning_factor(LEAF['thin_sr'], 0.0794, d), LEAF[f'thin_{k}'], rtol=1e-9)
    got = viscosity.shear_rate_avg(LEAF['sr_gx'], LEAF['sr_gy'], LEAF['sr_h'], 0.1, 0., 0.0794)
    np.testing.assert_allclose(got, LEAF['sr_out'], rtol=1e-9)
    # unknown law: viscosity unchanged (viscosity.py:63-64)
    np.testing.assert_array_equal(viscosity.piezoviscosity(LEAF['piezo_p'], 0.0794, {'name': 'none'}), np.full(20, 0.0794))
