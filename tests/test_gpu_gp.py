"""GP surrogate closure on the GPU against oracle/gp.py.

PARITY UNPINNED with respect to the reference's tinygp/jax stack (not installable here; the reference holds
no numeric GP fixtures): these tests pin the HIP kernels + rocSOLVER path to the NumPy/SciPy restatement of
the published Matern-3/2 / Cholesky formulas, and re-run the reference's own self-consistency test
(tests/test_inference.py:88-111)."""
import ctypes as C
import io
import os

import numpy as np
import pytest
from scipy.stats import qmc

from oracle import closures as ocl
from oracle import gp as ogp
from oracle.config import read_yaml_input as oracle_reader
from oracle.problem import OracleProblem

pytestmark = pytest.mark.gpu

SIM2D = """
options: {silent: True, write_freq: 1000}
grid: {Nx: 40, Ny: 24, Lx: 0.05, Ly: 0.03, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007,
       yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P']}
geometry: {type: inclined, hmax: 6.6e-5, hmin: 2.e-5, U: 20., V: 3.}
numerics: {CFL: 0.3, adaptive: 1, tol: 1.e-10, max_it: 100}
properties: {EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, C1: 3.5e9}
gp:
    press: {atol: 1., rtol: 0.1, obs_stddev: 1.e5, active_learning: False}
    shear: {atol: 1., rtol: 0.1, obs_stddev: 500., active_learning: False}
db: {init_size: 48, init_method: lhc, init_width: 0.001}
"""


def training_set(d, n, seed=5):
    """Mock-MD semantics (md/mock.py:81-107) evaluated with the oracle's closures: X (n,7), Y (n,13), Yerr (n,13)."""
    rng = np.random.default_rng(seed)
    prop, geo, gp = d['properties'], d['geometry'], d['gp']
    rho0 = prop['rho0']
    w = d['db']['init_width']
    X = np.column_stack([rng.uniform((1 - w) * rho0, (1 + w) * rho0, n),
                         rng.uniform(0.3, 0.9, n) * rho0 * geo['U'], rng.uniform(-0.3, 0.9, n) * rho0 * max(geo['V'], 1.),
                         rng.uniform(geo['hmin'], geo['hmax'], n), np.full(n, (geo['hmin'] - geo['hmax']) / d['grid']['Lx']),
                         np.zeros(n), np.zeros(n)])
    q, h = X[:, :3].T, X[:, 3:6].T
    bot = ocl.stress_bottom(q, h, geo['U'], geo['V'], prop['shear'], prop['bulk'], X[:, 6])
    top = ocl.stress_top(q, h, geo['U'], geo['V'], prop['shear'], prop['bulk'], X[:, 6])
    sp, ss = gp['press']['obs_stddev'], gp['shear']['obs_stddev']
    Y = np.column_stack([ocl.eos_pressure(X[:, 0], prop) + sp * rng.standard_normal(n), (bot + ss * rng.standard_normal((1, n))).T,
                         (top + ss * rng.standard_normal((1, n))).T])
    Ye = np.tile(np.array([sp, 0, 0, 0, ss, ss, 0, 0, 0, 0, ss, ss, 0.]), (n, 1))
    return X, Y, Ye


def build(sim=SIM2D, n=48):
    from gapflow_amd import Problem
    from gapflow_amd.io import read_yaml_input
    from gapflow_amd.gp import Database, Mock
    d = read_yaml_input(io.StringIO(sim))
    X, Y, Ye = training_set(d, n)
    db = Database(Mock(d['properties'], d['geometry'], d['gp']), d['db'])
    db.set_arrays(X, Y, Ye)
    prob = Problem(d['options'], d['grid'], d['numerics'], d['properties'], d['geometry'], gp=d['gp'], database=db)
    for m in prob._gp_models.values():
        m.optimise = False       # initial hyper-parameters: K stays well conditioned, so two Cholesky codes can be compared
    prob._pre_run()
    od = oracle_reader(io.StringIO(sim))
    ref = OracleProblem.from_dict(od)
    ref.gp_models = {}
    for name, kind in (('zz', 'press'), ('xz', 'shear_x'), ('yz', 'shear_y')):
        m = prob._gp_models.get(name)
        ref.gp_models[kind] = ogp.OracleSurrogate(kind, X, Y, Ye, m.theta, m.active_dims) if m is not None else None
    ref._pre_run()
    return prob, ref


def test_gp_fit_matches_oracle(hiplib):
    """Kernel-matrix fill + rocSOLVER dpotrf/dpotrs against SciPy on the same data."""
    from gapflow_amd import _lib
    rng = np.random.default_rng(3)
    for n, d, m in ((64, 2, 1), (200, 3, 2), (512, 3, 2)):
        X = rng.uniform(0.5, 1.0, (n, d))
        Y = np.column_stack([np.sin(4 * X[:, 0]) + X[:, -1]**2, np.cos(3 * X[:, 0]) - X[:, -1]])[:, :m] + 0.01 * rng.standard_normal((n, m))
        amp, inv_scale, sigma = 1.3, np.array([2.0, 0.7, 1.5])[:d], 0.05
        L, alpha, logdet = np.empty((n, n)), np.empty((n, m)), C.c_double(0)
        Xc, Yc, sc = _lib.f64c(X), _lib.f64c(Y), _lib.f64c(inv_scale)
        _lib.check(hiplib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), amp, _lib.as_dp(sc), sigma,
                                     _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
        ref = ogp.Fit(X, Y, amp, inv_scale, sigma)
        np.testing.assert_allclose(L, ref.L, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(alpha, ref.alpha, rtol=1e-7, atol=1e-9 * np.abs(ref.alpha).max())
        np.testing.assert_allclose(logdet.value, ref.logdet, rtol=1e-11)
        # the fit reproduces the data to the noise level and is what solving K alpha = Y means
        np.testing.assert_allclose(ref.K @ alpha, Y, atol=1e-9 * np.abs(Y).max())
    a = (C.c_double * 4)(1., 2., 1., 2.)    # duplicate inputs without noise: not positive definite -> error, not garbage
    assert hiplib.gpf_gp_fit(0, 2, 2, 1, a, a, 1.0, a, 0.0, None, None, None) == -4


def test_gp_fields_match_oracle(hiplib):
    """Posterior mean of p and of the wall shear stresses, predictive variances, GP sound speed."""
    prob, ref = build()
    prob.q[0] *= 1.0 + 2e-3 * np.sin(np.arange(prob.q[0].size).reshape(prob.q[0].shape) / 7.)
    prob.q[1] *= 1.0 + 0.1 * np.cos(np.arange(prob.q[0].size).reshape(prob.q[0].shape) / 5.)
    ref.q[...] = prob.q
    ref.update_closures(predictor=True, compute_var=True)
    scale = lambda a: np.abs(a).max()
    # agreement is limited by cond(K) * eps of two different Cholesky implementations, not by the kernels
    cond = np.linalg.cond(ref.gp_models['press'].fit.K)
    assert cond < 1e8, cond
    np.testing.assert_allclose(prob.pressure.pressure, ref.pressure, rtol=0, atol=1e-9 * scale(ref.pressure))
    lower = prob.wall_stress_xz.lower + prob.wall_stress_yz.lower
    upper = prob.wall_stress_xz.upper + prob.wall_stress_yz.upper
    for k in (3, 4):
        np.testing.assert_allclose(lower[k], ref.wall_lower[k], rtol=0, atol=1e-9 * scale(ref.wall_lower[k]))
        np.testing.assert_allclose(upper[k], ref.wall_upper[k], rtol=0, atol=1e-9 * scale(ref.wall_upper[k]))
    for name, kind in (('zz', 'press'), ('xz', 'shear_x'), ('yz', 'shear_y')):
        mean, var = prob._gp_models[name]._infer_mean_var()
        ovar = ref.gp_models[kind].variance
        # var = A - |L^-1 ks|^2 cancels to ~1e-3 A near the data: absolute agreement relative to A * Yscale^2
        tol = 1e-9 * ref.gp_models[kind].fit.amp * ref.gp_models[kind].Yscale**2
        np.testing.assert_allclose(var, ovar, rtol=0, atol=tol)
        np.testing.assert_allclose(prob._gp_models[name].maximum_variance, ovar.max(), rtol=0, atol=tol)
    np.testing.assert_allclose(prob.pressure.v_sound, ref.v_sound, rtol=1e-9)


def test_gp_steps_match_oracle(hiplib):
    """Five MacCormack steps with all three surrogates against the oracle with the same training data."""
    prob, ref = build()
    np.testing.assert_allclose(prob.dt, ref.dt, rtol=1e-10)
    for _ in range(5):
        prob.update()
        ref.update()
    for c in range(3):
        s = np.abs(ref.q[c]).max()
        assert np.abs(prob.q[c] - ref.q[c]).max() <= 2e-9 * s, f'component {c}'
    np.testing.assert_allclose(prob.dt, ref.dt, rtol=1e-9)
    np.testing.assert_allclose(prob.kinetic_energy, ref.kinetic_energy, rtol=1e-9)
    assert prob.step == ref.step == 5


def _pass_counts(prob):
    launched, reused = (C.c_int64 * 3)(), C.c_int64()
    assert prob._lib.gpf_gp_pass_counts(prob._h, launched, C.byref(reused)) == 0
    return list(launched), reused.value


def test_sound_speed_pass_serves_the_next_steps_first_pressure_evaluation(hiplib, monkeypatch):
    """The pass that closes a step (slope of the pressure surrogate's mean on the new state, stress.py:533-537) also keeps
    the mean; the next step's first closure update (same state, same model) copies it: six posterior-mean passes per step
    instead of seven, the same field bit for bit as with the reuse switched off, and no reuse across a model change."""
    prob, _ = build()
    monkeypatch.setenv('GPF_GP_NO_STATE_MEAN', '1')
    plain, _ = build()
    monkeypatch.delenv('GPF_GP_NO_STATE_MEAN')
    l0, r0 = _pass_counts(prob)
    p0, _ = _pass_counts(plain)
    nsteps = 6
    for _ in range(nsteps):
        prob.update()
        plain.update()
    l1, r1 = _pass_counts(prob)
    p1, pr = _pass_counts(plain)
    assert pr == 0 and p1[0] - p0[0] == 3 * nsteps and p1[1] - p0[1] == 2 * nsteps == p1[2] - p0[2]
    assert r1 - r0 >= nsteps - 1 and (l1[0] - l0[0]) + (r1 - r0) == 3 * nsteps      # pressure: 2 launched + 1 reused per step
    assert l1[1] - l0[1] == 2 * nsteps == l1[2] - l0[2]
    assert np.array_equal(prob.q, plain.q) and prob.dt == plain.dt
    # a refit in between (active learning does this) must not be served from the old model's mean
    for pb in (prob, plain):
        m = pb._gp_models['zz']
        m.theta = m.theta + 0.05
        m.attach()
    before, rb = _pass_counts(prob)
    prob.update()
    plain.update()
    after, ra = _pass_counts(prob)
    assert ra == rb and after[0] - before[0] == 3          # nothing reused in the step after the refit
    assert np.array_equal(prob.q, plain.q)
    prob.update()
    assert _pass_counts(prob)[1] == ra + 1                  # ... and reuse resumes with the step after


@pytest.mark.parametrize('ntrain', [5, 48, 200, 257, 512])
def test_fused_variance_kernel_matches_the_tiled_path(hiplib, monkeypatch, ntrain):
    """k_gp_var_fused (Matern values through LDS, L^-1 Ks on the f64 matrix cores, column norms in registers) against the
    tiled path (Ks tile in HBM, rocBLAS dgemm, norm kernel): same variance field to rounding -- both sum 48..512 squares
    per cell in different orders -- for all three surrogates (d = 2 and d = 3), row counts that fill 2, 7 and all 16 of the
    32-row blocks, and a cell count that is no multiple of the 64 cells of a workgroup."""
    prob, _ = build(n=ntrain)
    for name in ('zz', 'xz', 'yz'):
        m = prob._gp_models[name]
        monkeypatch.setenv('GPF_GP_VARIANCE', 'tiles')
        m.compute_variance(on_open_step=False)
        tiled, tiled_max = m.variance.copy(), m.maximum_variance
        monkeypatch.delenv('GPF_GP_VARIANCE')
        m.compute_variance(on_open_step=False)
        fused, fused_max = m.variance.copy(), m.maximum_variance
        scale = np.abs(tiled).max()
        assert np.isfinite(fused).all() and scale > 0
        # var = A - |v|^2 with |v|^2 <= A: the rounding of the sum is relative to A, not to the (possibly tiny) difference
        prior = m.kernel_variance * float(m.Yscale)**2
        assert np.abs(fused - tiled).max() <= 1e-11 * max(prior, scale), name
        np.testing.assert_allclose(fused_max, tiled_max, rtol=1e-9)


@pytest.mark.parametrize('n,d,m', [(64, 2, 1), (200, 3, 2), (512, 3, 2)])
def test_device_likelihood_matches_the_host_statement(hiplib, n, d, m):
    """gpf_gp_nll_eval (kernel matrix, Cholesky, K^-1 and the gradient sums on the device) against NegLogLikelihood (NumPy +
    LAPACK; itself compared with the oracle and with finite differences in tests/test_host_gp.py): value and gradient at
    three thetas around the starting point of a training, and a theta at which K is not positive definite."""
    from gapflow_amd.gp import NegLogLikelihood, DeviceNegLogLikelihood
    rng = np.random.default_rng(n + d)
    X = rng.uniform(-1., 1., (n, d))
    Y = np.column_stack([np.sin(2 * X[:, 0]) * (1 + 0.3 * X[:, -1]) + 0.01 * rng.standard_normal(n) for _ in range(m)])
    sigma = 0.02
    host = NegLogLikelihood(X, Y, sigma)
    theta0 = np.concatenate([[0.0], np.log(np.std(X, axis=0))])
    with DeviceNegLogLikelihood(X, Y, sigma) as dev:
        for shift in (0.0, 0.4, -0.7):
            theta = theta0 + shift * np.linspace(1., -1., 1 + d)
            fh, gh = host(theta)
            fd, gd = dev(theta)
            np.testing.assert_allclose(fd, fh, rtol=1e-10, atol=1e-8)
            np.testing.assert_allclose(gd, gh, rtol=1e-7, atol=1e-7 * np.abs(gh).max())
        bad = np.concatenate([[40.0], np.full(d, 30.0)])         # amplitude e^40, all points on top of each other: singular
        assert dev(bad)[0] == 1e300 == host(bad)[0]


def test_device_gradient_at_a_trained_ill_conditioned_theta(hiplib):
    """The likelihood gradient where training ends up: amplitude e^2 over a noise of 1e-5 of its square root and
    length scales of ten domain widths, cond(K) ~ 6e8 in the infinity norm.  Truth = the same formula in 40-digit arithmetic (mpmath, n = 48);
    the device gradient (K^-1 = Y Y^T with Y = L^-T by substitution, gp_inverse_transposed) must be as close to it as the
    LAPACK statement on the host (dpotrf / dpotri) is, up to a factor of 10."""
    import mpmath as mp
    from gapflow_amd.gp import NegLogLikelihood, DeviceNegLogLikelihood
    rng = np.random.default_rng(3)
    n, d, m = 48, 2, 1
    X = rng.uniform(-1., 1., (n, d))
    Y = (np.sin(2 * X[:, 0]) * (1 + 0.3 * X[:, 1]))[:, None]
    theta = np.array([2.0, 3.0, 3.4])
    sigma = 1e-5 * np.exp(theta[0] / 2)
    # --- 40 digits ---
    mp.mp.dps = 40
    amp, s = mp.e**mp.mpf(float(theta[0])), [mp.e**(-mp.mpf(float(t))) for t in theta[1:]]
    sq3 = mp.sqrt(3)
    Kf = mp.matrix(n, n)
    dK = [mp.matrix(n, n) for _ in range(d)]
    for i in range(n):
        for j in range(n):
            sd = [s[k] * (mp.mpf(float(X[i, k])) - mp.mpf(float(X[j, k]))) for k in range(d)]
            r = mp.sqrt(sum(v * v for v in sd))
            E = mp.e**(-sq3 * r)
            Kf[i, j] = amp * (1 + sq3 * r) * E
            for k in range(d):
                dK[k][i, j] = 3 * amp * E * sd[k]**2
    K = Kf.copy()
    for i in range(n):
        K[i, i] += mp.mpf(float(sigma))**2
    Kinv = K**-1
    y = mp.matrix([mp.mpf(float(v)) for v in Y[:, 0]])
    alpha = Kinv * y
    W = alpha * alpha.T - m * Kinv
    truth = np.array([float(-mp.mpf(1) / 2 * sum(W[i, j] * G[i, j] for i in range(n) for j in range(n))) for G in [Kf] + dK])
    cond = float(mp.norm(K, 'inf') * mp.norm(Kinv, 'inf'))
    assert cond > 1e8, cond
    gh = NegLogLikelihood(X, Y, sigma)(theta)[1]
    with DeviceNegLogLikelihood(X, Y, sigma) as dev:
        fd, gd = dev(theta)
    assert fd < 1e300, 'the device factorisation must succeed at this theta'
    eh, ed = np.abs(gh - truth), np.abs(gd - truth)
    print(f'\n[likelihood gradient, cond(K) = {cond:.1e}] truth {truth}; |host - truth| {eh}; |device - truth| {ed}')
    assert (ed <= 10 * eh + 1e-9 * np.abs(truth).max()).all(), (ed, eh)


def test_training_on_the_device_finds_the_host_optimum(hiplib, monkeypatch):
    """Surrogate.train with the objective on the device (default) and with GPF_GP_TRAIN=host: BFGS from the same start ends
    at the same objective value (1e-6 of it).  The hyper-parameters themselves agree only where the likelihood determines
    them: a length scale of e^16 (an input the pressure does not depend on) is a flat direction in which BFGS stops wherever
    rounding takes it."""
    from gapflow_amd.gp import NegLogLikelihood
    thetas = {}
    for mode in ('device', 'host'):
        monkeypatch.setenv('GPF_GP_TRAIN', mode)
        prob, _ = build(n=96)
        m = prob._gp_models['zz']
        m.train(reason=0, optimise=True)
        thetas[mode] = m.theta.copy()
    nll = NegLogLikelihood(m.Xtrain, m.Ytrain, m.Yerr)
    fd, fh = nll(thetas['device'])[0], nll(thetas['host'])[0]
    assert abs(fd - fh) <= 5e-6 * abs(fh), (fd, fh, thetas)      # (BFGS stops at |gradient| < 1e-5 on either objective)
    well = np.exp(-thetas['host'][1:]) > 1e-3            # inverse length scales that matter
    np.testing.assert_allclose(thetas['device'][1:][well], thetas['host'][1:][well], atol=0.05)


def test_blocked_cholesky_factorises_a_numerically_singular_trained_kernel_matrix(hiplib):
    """What training produces on a narrow Latin-hypercube sample: amplitude e^11.7 over noise 6e-5 -- the smallest eigenvalue of
    K computes to -2e-9, LAPACK's dpotrf still succeeds (smallest pivot ~ sigma).  The blocked device factorisation must too
    (its panel solve is a substitution kernel for that reason: through rocBLAS dtrsm's inverted diagonal blocks it reported a
    non-positive pivot here and the run died in gpf_gp_set_model), with the same factor to 1e-4 of sigma-sized entries."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scipy.linalg import lapack
    from bench import GP_YAML
    from gapflow_amd import Problem, _lib
    text = GP_YAML.format(n=64, nt=256).replace('obs_stddev: 100., active_learning: False', 'obs_stddev: 1.e5, active_learning: False')
    prob = Problem.from_string(text)
    for m in prob._gp_models.values():
        m.optimise = False
    prob._pre_run()
    m = prob._gp_models['zz']
    X, Y, s = m.Xtrain, m.Ytrain, m.Yerr
    th = np.array([11.70824469, 2.93533174, 14.77329233])          # the optimum BFGS finds for this training set
    amp, inv = np.exp(th[0]), np.exp(-th[1:])
    Z = X * inv
    r = np.sqrt(3 * ((Z[:, None, :] - Z[None, :, :])**2).sum(-1))
    K = amp * (1 + r) * np.exp(-r) + s**2 * np.eye(len(X))
    assert np.linalg.eigvalsh(K)[0] < 1e-6 * s**2 + 1e-8             # numerically singular
    c, info = lapack.dpotrf(K, lower=True)
    assert info == 0
    L, alpha, ld = np.zeros_like(K), np.zeros((len(X), 1)), C.c_double()
    lib = _lib.require_device()
    rc = lib.gpf_gp_fit(0, len(X), X.shape[1], 1, _lib.as_dp(_lib.f64c(X)), _lib.as_dp(_lib.f64c(Y)), amp, _lib.as_dp(_lib.f64c(inv)), s,
                        _lib.as_dp(L), _lib.as_dp(alpha), C.byref(ld))
    assert rc == 0, lib.gpf_last_error().decode()
    assert np.diag(L).min() > 0.5 * s
    assert np.abs(L - np.tril(c)).max() <= 1e-4 * np.abs(c).max()
    # and the model attaches: mean and variance evaluate
    m.theta = th
    m.attach()
    m.compute_variance(on_open_step=False)
    assert np.isfinite(m.maximum_variance)


def test_predict_repredict_self_consistency(hiplib):
    """tests/test_inference.py:88-111 of the reference: a fresh prediction equals the cached re-prediction."""
    prob, _ = build()
    for _ in range(3):
        p1, v1 = prob.pressure._infer_mean_var()
        s1, w1 = prob.wall_stress_xz._infer_mean_var()
        p2, v2 = prob.pressure._infer_mean_var()
        s2, w2 = prob.wall_stress_xz._infer_mean_var()
        assert np.isclose(np.max(np.abs(p1 - p2)), 0.) and np.isclose(np.max(np.abs(v1 - v2)), 0.)
        assert np.isclose(np.max(np.abs(s1 - s2)), 0.) and np.isclose(np.max(np.abs(w1 - w2)), 0.)
        prob.update()


def test_active_learning_from_yaml(hiplib):
    """The reference's inference test set-up (1-D parabolic slider, BWR EOS, Mock MD, active learning): the
    database grows where the variance criterion asks for it and the run stays valid."""
    from gapflow_amd import Problem
    sim = """
options: {silent: True, write_freq: 100}
grid: {Lx: 1470., Ly: 1., Nx: 200, Ny: 1, xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], yS: ['P', 'P', 'P'], yN: ['P', 'P', 'P'],
       xE_D: 0.8, xW_D: 0.8}
geometry: {type: parabolic, hmin: 12., hmax: 60., U: 0.12, V: 0.}
numerics: {CFL: 0.5, adaptive: 1, tol: 1e-8, dt: 0.05, max_it: 5000}
properties: {shear: 2.15, bulk: 0., EOS: BWR, T: 1.0, rho0: 0.8}
gp:
    press: {fix_noise: True, atol: .7, rtol: 0., obs_stddev: 2.e-2, max_steps: 10, active_learning: True}
    shear: {fix_noise: True, atol: .9, rtol: 0., obs_stddev: 4.e-3, max_steps: 10, active_learning: True}
db: {init_size: 3, init_method: rand, init_width: 0.01}
"""
    prob = Problem.from_string(sim)
    prob._pre_run()
    n0 = prob.database.size
    assert n0 == 3
    for _ in range(3):
        prob.update()
    assert prob.step == 3 and not prob._stop
    assert prob.database.size > n0
    assert np.isfinite(prob.q).all() and (prob.q[0] > 0).all()
    for m in prob._gp_models.values():
        assert m.last_fit_train_size == prob.database.size or m._pause >= 0


@pytest.mark.parametrize('method', ['rand', 'lhc', 'sobol'])
def test_database_addition_and_reload(hiplib, tmp_path, method):
    """tests/test_database.py:30-55 of the reference: initialise, add, and find everything again through a second
    Database on the same dtool_path."""
    from gapflow_amd import Database
    from gapflow_amd.md import Mock
    db_config = {'init_size': 4, 'init_width': 0.01, 'init_method': method, 'init_seed': 42, 'dtool_path': str(tmp_path)}
    geo = {'U': 1., 'V': 0.}
    prop = {'shear': 1., 'bulk': 0., 'EOS': 'PL'}
    gp = {'press_gp': False, 'shear_gp': False}
    md = Mock(prop, geo, gp)
    db = Database(md, db_config, num_extra_features=1)
    Xtest = np.random.default_rng(0).uniform(0.1, 1., size=(100, 7))
    db.initialize(Xtest)
    assert db.size == db_config['init_size']
    Xnew = np.random.default_rng(1).uniform(0.1, 1., size=(10, 7))
    db.add_data(Xnew)
    assert db.size == 14
    new_db = Database(md, db_config, num_extra_features=1)
    assert new_db.size == 14
    np.testing.assert_allclose(np.sort(new_db._Xtrain, axis=0), np.sort(db._Xtrain, axis=0), rtol=1e-15)
    np.testing.assert_allclose(np.sort(new_db._Ytrain, axis=0), np.sort(db._Ytrain, axis=0), rtol=1e-15)
    # Mock semantics (md/mock.py:92-96): PL pressure with the function's defaults, noise-free here
    from oracle import closures as ocl
    p_ref = ocl.eos_pressure(db._Xtrain[:, 0], {'EOS': 'PL', 'rho0': 1.1853, 'P0': 101325., 'alpha': 0.})
    np.testing.assert_allclose(db._Ytrain[:, 0], p_ref, rtol=1e-12)


def test_variance_fields_are_written_with_the_frames(hiplib, tmp_path):
    """problem.py:196-203: with surrogates sol.nc also carries pressure_var / wall_stress_xz_var (/ _yz_var in 2-D), the
    fields viz/plotting.py:340-348 reads for its uncertainty bands; refreshed one step before each frame (problem.py:530)."""
    from scipy.io import netcdf_file
    from gapflow_amd import Problem
    sim = SIM2D.replace("options: {silent: True, write_freq: 1000}",
                        f"options: {{output: {tmp_path}/run, write_freq: 4, use_tstamp: False, silent: False}}").replace('max_it: 100', 'max_it: 8')
    prob = Problem.from_string(sim)
    for m in prob._gp_models.values():
        m.optimise = False
    prob.run()
    with netcdf_file(os.path.join(tmp_path, 'run', 'sol.nc'), 'r', mmap=False) as f:
        nfr = f.variables['pressure'].shape[0]
        assert nfr == 3                                                     # steps 0, 4, 8
        for name, model in (('pressure_var', 'zz'), ('wall_stress_xz_var', 'xz'), ('wall_stress_yz_var', 'yz')):
            v = f.variables[name]
            assert v.shape == (nfr, 42, 26)
            last = v[-1]
            assert np.isfinite(last).all() and last.max() > 0.0 and last.min() > -1e-9 * last.max()
            np.testing.assert_array_equal(last, prob._gp_models[model].variance)


def test_surrogates_reproduce_the_reference_laws_within_their_own_sigma(hiplib):
    """SURVEY.md 8(c), the analytic GP check: surrogates trained on dense, noise-free samples of the fixed-form laws must
    give those laws back at unseen points to within their own predictive standard deviation -- on the device path (kernel
    matrix, Cholesky, posterior mean, variance tiles), with the hyper-parameters the host optimiser finds.  The truth at
    the test points is reference output: the Dowson-Higginson pressures of tests/golden/leaf_closures.npz (written by the
    reference's pressure.py) and the oracle's wall stresses (pinned to the reference's viscous.py by
    tests/test_oracle_golden.py).  This needs no tinygp: it checks the GP closure against the physics it stands in for."""
    from gapflow_amd import Problem
    from gapflow_amd.io import read_yaml_input
    from gapflow_amd.gp import Database, Mock
    from helpers import GOLDEN
    leaf = np.load(GOLDEN + '/leaf_closures.npz')
    rho_ref, p_ref = leaf['eos_DH_rho'].ravel(), leaf['eos_DH_p'].ravel()
    keep = rho_ref <= 1000.                             # below the law's clamp at 0.99 C2 rho0 (a kink no smooth GP fits)
    rho_ref, p_ref = rho_ref[keep], p_ref[keep]
    ncell = 48
    sim = f"""
options: {{silent: True, write_freq: 1000}}
grid: {{Nx: {ncell - 2}, Ny: 1, Lx: 0.05, Ly: 1., xE: ['D', 'N', 'N'], xW: ['D', 'N', 'N'], xE_D: 877.7007, xW_D: 877.7007}}
geometry: {{type: inclined, hmax: 3.e-5, hmin: 1.e-5, U: 0.1, V: 0.}}
numerics: {{CFL: 0.3, adaptive: 1, tol: 1.e-10, max_it: 100}}
properties: {{EOS: DH, shear: 0.0794, bulk: 0., rho0: 877.7007, P0: 101325., C1: 3.5e10, C2: 1.23}}
gp:
    press: {{atol: 1., rtol: 0.1, obs_stddev: 5.e6, active_learning: False}}
    shear: {{atol: 1., rtol: 0.1, obs_stddev: 0.1, active_learning: False}}
db: {{init_size: 400, init_method: lhc, init_width: 0.001}}
"""
    d = read_yaml_input(io.StringIO(sim))
    prop, geo = d['properties'], d['geometry']
    # training data: 400 space-filling, NOISE-FREE samples of the laws over the box the test points live in
    n = 400
    u = qmc.LatinHypercube(d=3, seed=5).random(n)
    X = np.zeros((n, 7))
    X[:, 0] = 800. + 205. * u[:, 0]
    X[:, 1] = 20. + 45. * u[:, 1]
    X[:, 3] = 1e-5 + 2e-5 * u[:, 2]
    X[:, 4] = (geo['hmin'] - geo['hmax']) / d['grid']['Lx']
    qx, hx = X[:, :3].T, X[:, 3:6].T
    bot = ocl.stress_bottom(qx, hx, geo['U'], geo['V'], prop['shear'], prop['bulk'], X[:, 6])
    top = ocl.stress_top(qx, hx, geo['U'], geo['V'], prop['shear'], prop['bulk'], X[:, 6])
    Y = np.column_stack([ocl.eos_pressure(X[:, 0], prop), bot.T, top.T])
    # the laws are sampled WITHOUT noise; obs_stddev is the jitter that keeps K factorisable (~1e-4 of the output scales:
    # the pressure depends on one of its two inputs only, so the fitted kernel is nearly degenerate along the other)
    sp, ss = d['gp']['press']['obs_stddev'], d['gp']['shear']['obs_stddev']
    Ye = np.tile(np.array([sp, 0, 0, 0, ss, ss, 0, 0, 0, 0, ss, ss, 0.]), (n, 1))
    db = Database(Mock(prop, geo, d['gp']), d['db'])
    db.set_arrays(X, Y, Ye)
    prob = Problem(d['options'], d['grid'], d['numerics'], prop, geo, gp=d['gp'], database=db)
    prob._pre_run()                                     # trains (host BFGS on the marginal likelihood) and factorises
    # test points: the reference's densities, fluxes and gaps strictly inside the training box
    rng = np.random.default_rng(9)
    q = prob.q
    m = min(len(rho_ref), ncell)
    q[0, :m, :] = rho_ref[:m, None]
    q[0, m:, :] = 900.
    q[1] = rng.uniform(25., 60., (ncell, 1))
    q[2] = 0.
    topo = prob.topo.full
    topo[0] = rng.uniform(1.2e-5, 2.8e-5, (ncell, 1))
    prob._upload_topo()
    p_mean, p_var = prob.pressure._infer_mean_var()
    s_mean, s_var = prob.wall_stress_xz._infer_mean_var()
    # (1) pressure against the reference's own outputs
    err = np.abs(p_mean[:m, 1] - p_ref[:m])
    sigma = np.sqrt(np.maximum(p_var[:m, 1], 0.))
    scale = np.abs(p_ref[:m]).max()
    assert (err <= 3.0 * sigma + 1e-9 * scale).all(), (err / scale).max()
    assert err.max() <= 1e-3 * scale, "the surrogate did not learn the law"
    # (2) wall shear (lower / upper wall share one variance: a two-output GP with one kernel matrix, stress.py:205-215)
    qc, hc = prob.q[:, :, 1], prob.topo.full[:3, :, 1]
    tb = ocl.stress_bottom(qc, hc, geo['U'], geo['V'], prop['shear'], prop['bulk'], np.zeros(ncell))[4]
    tt = ocl.stress_top(qc, hc, geo['U'], geo['V'], prop['shear'], prop['bulk'], np.zeros(ncell))[4]
    sig = np.sqrt(np.maximum(s_var[:, 1], 0.))
    scale = max(np.abs(tb).max(), np.abs(tt).max())
    for mean, truth in ((s_mean[0][:, 1], tb), (s_mean[1][:, 1], tt)):
        e = np.abs(mean - truth)
        assert (e <= 3.0 * sig + 1e-9 * scale).all(), (e / scale).max()
        assert e.max() <= 5e-3 * scale, "the surrogate did not learn the law"


def test_rocsolver_factorisation_equals_the_in_library_one(hiplib, tmp_path):
    """BASELINE.json's north star names 'a rocSOLVER Cholesky': gpf_gp_fit / gpf_gp_set_model / gpf_gp_nll_eval factorise with
    rocsolver_dpotrf / dpotrs (the copy beside the rocBLAS in use, mapped before the first HIP call: csrc/gp_kernels.hip roclibs(),
    _lib._preload_rocsolver; the hang and the abort of round 2 were a rocSOLVER of ROCm 7.2 meeting the HIP runtime and rocBLAS of
    the ROCm 7.0 that PyTorch bundles, profiles/r03_rocsolver/).  GPF_USE_ROCSOLVER=0 selects the in-library blocked Cholesky;
    the switch is read once per process: a child process runs that one, this process the default, and L, alpha and log det K
    must agree (and match SciPy)."""
    import subprocess
    import sys
    from scipy.linalg import lapack
    from gapflow_amd import _lib
    rng = np.random.default_rng(4)
    n, dd, m = 300, 3, 2
    X = rng.uniform(0.5, 1.0, (n, dd))
    Y = np.column_stack([np.sin(4 * X[:, 0]) + X[:, 2]**2, np.cos(3 * X[:, 1])]) + 0.01 * rng.standard_normal((n, m))
    amp, inv_scale, sigma = 1.2, np.array([2.0, 0.8, 1.5]), 0.05
    # second case: the numerically singular kernel matrix of a trained surrogate (see the test above)
    Xs, Ys, ss = singular_training_set()
    ths = np.array([11.70824469, 2.93533174, 14.77329233])
    np.savez(tmp_path / 'in.npz', X=X, Y=Y, inv_scale=inv_scale, Xs=Xs, Ys=Ys, ss=ss, ths=ths)
    code = f"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
from gapflow_amd import _lib
lib = _lib.require_device()
z = np.load({repr(str(tmp_path / 'in.npz'))})
out = dict(which=lib.gpf_gp_factorisation().decode())
for tag, X, Y, amp, inv, sigma in (('', z['X'], z['Y'], {amp}, z['inv_scale'], {sigma}), ('s', z['Xs'], z['Ys'], float(np.exp(z['ths'][0])), np.exp(-z['ths'][1:]), float(z['ss']))):
    X, Y, sc = _lib.f64c(X), _lib.f64c(Y), _lib.f64c(inv)
    n, d = X.shape; m = Y.shape[1]
    L, alpha, logdet = np.zeros((n, n)), np.zeros((n, m)), C.c_double(0)
    _lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(X), _lib.as_dp(Y), amp, _lib.as_dp(sc), sigma, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
    out['L' + tag], out['alpha' + tag], out['logdet' + tag] = L, alpha, logdet.value
np.savez({repr(str(tmp_path / 'out.npz'))}, **out)
"""
    env = dict(os.environ, GPF_USE_ROCSOLVER='0')
    res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    z = np.load(tmp_path / 'out.npz')
    assert str(z['which']) == 'in-library blocked Cholesky'
    assert hiplib.gpf_gp_factorisation().decode() == 'rocsolver_dpotrf', 'rocSOLVER must be what factorises by default on this image'

    def fit(X, Y, amp, inv, s):
        L, alpha, logdet = np.zeros((len(X), len(X))), np.zeros((len(X), Y.shape[1])), C.c_double(0)
        _lib.check(hiplib.gpf_gp_fit(0, len(X), X.shape[1], Y.shape[1], _lib.as_dp(_lib.f64c(X)), _lib.as_dp(_lib.f64c(Y)), amp, _lib.as_dp(_lib.f64c(inv)), s,
                                     _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
        return dict(L=L, alpha=alpha, logdet=logdet.value)
    mine = fit(X, Y, amp, inv_scale, sigma)
    ref = ogp.Fit(X, Y, amp, inv_scale, sigma)
    for name, got in (('in-library', z), ('rocSOLVER', mine)):
        np.testing.assert_allclose(np.tril(got['L']), ref.L, rtol=1e-9, atol=1e-12, err_msg=name)
        np.testing.assert_allclose(got['alpha'], ref.alpha, rtol=1e-7, atol=1e-9 * np.abs(ref.alpha).max(), err_msg=name)
        np.testing.assert_allclose(got['logdet'], ref.logdet, rtol=1e-11, err_msg=name)
    np.testing.assert_allclose(np.tril(z['L']), np.tril(mine['L']), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(z['alpha'], mine['alpha'], rtol=1e-8, atol=1e-10 * np.abs(mine['alpha']).max())
    # the singular one: both must succeed where LAPACK does, with its factor (1e-4 of the largest entry; sigma-sized pivots)
    amps, invs = float(np.exp(ths[0])), np.exp(-ths[1:])
    Zs = Xs * invs
    r = np.sqrt(3 * ((Zs[:, None, :] - Zs[None, :, :])**2).sum(-1))
    K = amps * (1 + r) * np.exp(-r) + ss**2 * np.eye(len(Xs))
    c, info = lapack.dpotrf(K, lower=True)
    assert info == 0
    sing = fit(Xs, Ys, amps, invs, ss)
    for name, Lg in (('in-library', z['Ls']), ('rocSOLVER', sing['L'])):
        e = np.abs(np.tril(Lg) - np.tril(c)).max() / np.abs(c).max()
        assert e <= 1e-4 and np.diag(Lg).min() > 0.5 * ss, f'{name}: {e:.2e}'
        print(f'[singular trained K, {name}] max |L - LAPACK| / max |L| = {e:.1e}, smallest pivot / sigma = {np.diag(Lg).min() / ss:.3f}')


def singular_training_set():
    """Training set of the pressure surrogate of a 64^2 slider with 256 Latin-hypercube points in a narrow box (bench.GP_YAML)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import GP_YAML
    from gapflow_amd import Problem
    text = GP_YAML.format(n=64, nt=256).replace('obs_stddev: 100., active_learning: False', 'obs_stddev: 1.e5, active_learning: False')
    prob = Problem.from_string(text)
    for m in prob._gp_models.values():
        m.optimise = False
    prob._pre_run()
    m = prob._gp_models['zz']
    return np.array(m.Xtrain), np.array(m.Ytrain), float(m.Yerr)
