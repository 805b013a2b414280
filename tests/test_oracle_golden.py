"""The oracle against the committed golden vectors (true outputs of the reference's leaf modules,
tests/golden/make_golden.py).  Runs on CPU; guards the checker itself."""
import numpy as np
import pytest

from helpers import STEP_CASES, load_case, input_dict, GOLDEN
from oracle import closures as ocl
from oracle import integrate as oint
from oracle.config import read_yaml_input
from oracle.problem import OracleProblem

LEAF = np.load(GOLDEN + '/leaf_closures.npz')
EOS_PROPS = {
    'DH': dict(EOS='DH', rho0=877.7007, P0=101325., C1=3.5e10, C2=1.23),
    'PL': dict(EOS='PL', rho0=1.1853, P0=101325., alpha=0.),
    'vdW': dict(EOS='vdW', M=39.948, T=100., a=1.355, b=0.03201),
    'MT': dict(EOS='MT', rho0=700., P0=0.101e6, K=0.557e9, n=7.33),
    'cubic': dict(EOS='cubic', a=1.33030e-1, b=-1.41778e2, c=8.35134e4, d=-2.86532e6),
    'BWR': dict(EOS='BWR', T=1.0, gamma=3.0),
    'Bayada': dict(EOS='Bayada', rho_l=850., rho_v=0.019, c_l=1600., c_v=352.),
}


@pytest.mark.parametrize('eos', sorted(EOS_PROPS))
def test_eos_and_sound_speed(eos):
    rho = LEAF[f'eos_{eos}_rho']
    np.testing.assert_allclose(ocl.eos_pressure(rho, EOS_PROPS[eos]), LEAF[f'eos_{eos}_p'], rtol=1e-13)
    with np.errstate(invalid='ignore'):
        np.testing.assert_allclose(ocl.eos_sound_speed(rho, EOS_PROPS[eos]), LEAF[f'eos_{eos}_c'], rtol=1e-13, equal_nan=True)


@pytest.mark.parametrize('tag', ['Ls0', 'LsF'])
def test_viscous_stresses(tag):
    q, h, Ls = LEAF['visc_q'], LEAF['visc_h'], LEAF[f'visc_{tag}_Ls'][0]
    U, V, eta, zeta = LEAF['visc_params']
    for fn, key in ((ocl.stress_bottom, 'bot'), (ocl.stress_top, 'top'), (ocl.stress_avg, 'avg')):
        ref = LEAF[f'visc_{tag}_{key}']
        got = fn(q, h, U, V, eta, zeta, Ls)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14 * np.abs(ref).max())


def test_flux_and_source():
    q, p, tau = LEAF['visc_q'], LEAF['flux_p'], LEAF['flux_tau']
    for d in (1, -1):
        fx, fy = oint.predictor_corrector(q, p, tau, d)
        np.testing.assert_array_equal(fx, LEAF[f'flux_d{d:+d}_x'])
        np.testing.assert_array_equal(fy, LEAF[f'flux_d{d:+d}_y'])
    out = oint.source(q, LEAF['src_topo'], tau, LEAF['src_lower'], LEAF['src_upper'])
    np.testing.assert_allclose(out, LEAF['src_out'], rtol=1e-15)


def test_viscosity_laws():
    pz = {'Barus': dict(name='Barus', aB=20e-9), 'Roelands': dict(name='Roelands', mu_inf=1e-3, p_ref=1.96e8, z=0.68),
          'Dukler': dict(name='Dukler', eta_v=3.9e-5, rho_l=850., rho_v=0.019),
          'McAdams': dict(name='McAdams', eta_v=3.9e-5, rho_l=850., rho_v=0.019)}
    for k, d in pz.items():
        arg = LEAF['piezo_rho'] if k in ('Dukler', 'McAdams') else LEAF['piezo_p']
        np.testing.assert_allclose(ocl.piezoviscosity(arg, 0.0794, d), LEAF[f'piezo_{k}'], rtol=1e-14)
    th = {'Eyring': dict(name='Eyring', tauE=5e5), 'Carreau': dict(name='Carreau', mu_inf=1e-9, lam=1e-6, a=2., N=0.6)}
    for k, d in th.items():
        np.testing.assert_allclose(ocl.shear_thinning_factor(LEAF['thin_sr'], 0.0794, d), LEAF[f'thin_{k}'], rtol=1e-14)
    np.testing.assert_allclose(ocl.shear_rate_avg(LEAF['sr_gx'], LEAF['sr_gy'], LEAF['sr_h'], 0.1, 0., 0.0794),
                               LEAF['sr_out'], rtol=1e-14)


@pytest.mark.parametrize('name', STEP_CASES)
def test_step_fixtures(name):
    """Step-level goldens (reference leaf arithmetic driven in problem.py order) are reproduced bit for bit."""
    fx, yaml_text, meta = load_case(name)
    prob = OracleProblem.from_dict(input_dict(yaml_text, meta, read_yaml_input))
    if meta.get('slip'):
        prob.extra[...] = fx['extra']
    prob._pre_run()
    if meta.get('wave'):
        prob.q[...] = fx['q_init']
        prob.kinetic_energy_old = prob.kinetic_energy
    np.testing.assert_array_equal(prob.topo, fx['topo'])
    snaps = sorted(meta['snaps'])
    for s in range(1, snaps[-1] + 1):
        prob.update()
        if s in snaps:
            np.testing.assert_allclose(prob.q, fx[f'q_{s}'], rtol=1e-12, atol=0)
            row = fx['history'][s - 1]
            np.testing.assert_allclose([prob.simtime, prob.dt, prob.kinetic_energy, prob.v_sound], row[[1, 2, 3, 5]], rtol=1e-12)
