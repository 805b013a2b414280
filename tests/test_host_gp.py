"""Hyper-parameter training objective of the GP closures (reference: GaPFlow/models/gp.py:290-335,
576-603) on the host: gapflow_amd.gp.NegLogLikelihood (cached squared differences, LAPACK dpotri, hand-derived
gradient) against the oracle's plain restatement (oracle/gp.py:73-104) and against central finite differences, and
the trained hyper-parameters of both optimiser runs on one 64-point set.

PARITY UNPINNED with respect to tinygp / jaxopt (not installable here): what is pinned is that the product's
objective IS -sum_o log N(Y_o | 0, K(theta)) of the Matern-3/2 ARD kernel the reference builds, and that its
gradient is the gradient of that function."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import gp as ogp


def data(n, d, m, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.4, 1.0, (n, d))
    Y = np.column_stack([np.sin(5 * X[:, 0]) + X[:, -1]**2, np.cos(3 * X[:, 0]) - 0.5 * X[:, -1]])[:, :m]
    return X, Y + 0.02 * rng.standard_normal((n, m))


THETAS = {2: [np.array([0.0, np.log(0.2), np.log(0.25)]), np.array([1.3, -0.4, 0.6]), np.array([-2.0, -2.5, 0.1])],
          3: [np.array([0.0, np.log(0.2), np.log(0.3), np.log(0.25)]), np.array([0.7, 0.3, -1.1, 0.5]), np.array([-1.0, -2.0, 1.0, -0.5])]}


@pytest.mark.parametrize('n,d,m', [(40, 2, 1), (64, 3, 2), (150, 3, 2)])
def test_objective_and_gradient_match_oracle_and_finite_differences(n, d, m):
    from gapflow_amd.gp import NegLogLikelihood, neg_log_likelihood
    X, Y = data(n, d, m, seed=n)
    sigma = 0.03
    nll = NegLogLikelihood(X, Y, sigma)
    for theta in THETAS[d]:
        f, g = nll(theta)
        fo, go = ogp.neg_log_likelihood(theta, X, Y, sigma)
        assert f == pytest.approx(fo, rel=1e-11, abs=1e-9)
        np.testing.assert_allclose(g, go, rtol=1e-8, atol=1e-8 * np.abs(go).max())
        f2, g2 = neg_log_likelihood(theta, X, Y, sigma)          # the functional form used by Surrogate.train
        assert f2 == f and np.array_equal(g2, g)
        # central differences of the ORACLE's value: the product's gradient is the gradient of that function
        fd = np.empty_like(theta)
        for k in range(len(theta)):
            e = np.zeros_like(theta)
            e[k] = 1e-5
            fd[k] = (ogp.neg_log_likelihood(theta + e, X, Y, sigma)[0] - ogp.neg_log_likelihood(theta - e, X, Y, sigma)[0]) / 2e-5
        np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-6 * np.abs(fd).max())
    # the value is the Gaussian log-density it claims to be (direct evaluation through numpy.linalg)
    theta = THETAS[d][0]
    K = ogp.matern32(X, X, np.exp(theta[0]), np.exp(-theta[1:])) + sigma**2 * np.eye(n)
    sign, logdet = np.linalg.slogdet(K)
    direct = sum(0.5 * Y[:, o] @ np.linalg.solve(K, Y[:, o]) + 0.5 * logdet + 0.5 * n * np.log(2 * np.pi) for o in range(m))
    assert sign > 0 and nll(theta)[0] == pytest.approx(direct, rel=1e-10)


def test_objective_rejects_unusable_probes():
    """A line-search probe far outside the sensible range must come back as 'worse', not as NaN (gp.py:320-321 lets
    SciPy's BFGS probe freely)."""
    from gapflow_amd.gp import NegLogLikelihood
    X, Y = data(30, 2, 1, seed=1)
    f, g = NegLogLikelihood(X, Y, 0.0)(np.array([800.0, 0.0, 0.0]))       # exp overflow
    assert f == 1e300 and not g.any()
    Xd = np.vstack([X, X[:1]])                                            # duplicate input, no noise: K singular
    f, g = NegLogLikelihood(Xd, np.vstack([Y, Y[:1]]), 0.0)(np.zeros(3))
    assert f == 1e300 and not g.any()


def test_trained_hyperparameters_agree_with_oracle():
    """The same BFGS on both objectives from the reference's initial guess (log_amp 0, log_scale = log std X,
    stress.py:281-284) ends at the same optimum."""
    from gapflow_amd.gp import NegLogLikelihood
    X, Y = data(64, 3, 2, seed=7)
    sigma = 0.02
    theta0 = ogp.OracleSurrogate.theta_init(X)
    res = minimize(NegLogLikelihood(X, Y, sigma), theta0, jac=True, method='BFGS')
    theta_o, f_o = ogp.train(X, Y, sigma, theta0)
    assert res.fun == pytest.approx(f_o, rel=1e-8)
    np.testing.assert_allclose(res.x, theta_o, rtol=1e-4, atol=1e-4)
    # ... and it is a stationary point of the oracle's objective
    g = ogp.neg_log_likelihood(res.x, X, Y, sigma)[1]
    assert np.abs(g).max() < 1e-3 * max(1.0, abs(f_o))
    # training improved on the initial guess
    assert res.fun < ogp.neg_log_likelihood(theta0, X, Y, sigma)[0]
