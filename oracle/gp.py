"""Gaussian-process surrogate closure, restated in NumPy/SciPy.  Test infrastructure only.

PARITY UNPINNED.  The reference builds its GP with third-party tinygp (>= 0.3.0, no upper pin), jax
(<= 0.9.0) and jaxopt (== 0.8.5) (pyproject.toml:36-41); none of them is installed here and the
reference's only numeric GP test is a self-consistency check (tests/test_inference.py:88-111,
"fresh predict == cached re-predict"), which tests/ re-runs.  What is restated below is the published
algorithm behind the reference's call sites:

  GaPFlow/models/gp.py:576-603   kernel = exp(log_amp) * Linear(exp(-log_scale), Matern32(L2Distance))
                                 k(x,x') = A (1 + sqrt3 r) exp(-sqrt3 r),  r = || s o (x - x') ||_2
                                 GaussianProcess(kernel, X, diag=yerr**2):  K = k(X,X) + yerr^2 I = L L^T
  gp.py:307-318                  loss = - sum_outputs log N(Y_o | 0, K)
  gp.py:538-570                  condition(): alpha = K^-1 y, mean = Ks^T alpha, noise on test points = 0
  gp.py:509-535                  re-predict: mean = Ks^T alpha; v = L^-1 Ks; var = k(x*,x*) [= A] + 0 - sum v^2
  models/stress.py:533-537       v_sound^2 = max_cells d mean / d x_0 * Yscale / X_scale[0]
                                 (closed form: dk/dx_0 = -3 A s_0^2 (x_0 - x_0') exp(-sqrt3 r))
  gp.py:223-232, stress.py:195-197, 542-544   test inputs: all cells incl. ghosts, features
                                 [rho, jx, jy, h, dh/dx, dh/dy, extra] / X_scale, then active dims

tinygp behaviours assumed from its documentation (unverifiable offline): a kernel called with one
argument returns its diagonal; `condition` adds no noise to the test points; Linear(scale) multiplies
the inputs elementwise; L2Distance is the Euclidean norm.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve, solve_triangular
from scipy.optimize import minimize

SQRT3 = np.sqrt(3.0)


def matern32(X1, X2, amp, inv_scale):
    """k(X1, X2) of shape (n1, n2) for inputs (n, d)."""
    d = (X1[:, None, :] - X2[None, :, :]) * inv_scale
    r = np.sqrt(np.sum(d * d, axis=-1))
    return amp * (1.0 + SQRT3 * r) * np.exp(-SQRT3 * r)


class Fit:
    """K = k(X,X) + sigma^2 I = L L^T and alpha = K^-1 Y for fixed hyper-parameters."""

    def __init__(self, X, Y, amp, inv_scale, sigma):
        self.X = np.asarray(X, float)
        self.Y = np.asarray(Y, float).reshape(len(self.X), -1)
        self.amp, self.inv_scale, self.sigma = float(amp), np.asarray(inv_scale, float), float(sigma)
        K = matern32(self.X, self.X, self.amp, self.inv_scale) + self.sigma**2 * np.eye(len(self.X))
        self.K = K
        self.L = np.linalg.cholesky(K)
        self.alpha = cho_solve((self.L, True), self.Y)
        self.logdet = 2.0 * np.sum(np.log(np.diag(self.L)))

    def log_probability(self):
        n = len(self.X)
        return np.array([-0.5 * self.Y[:, o] @ self.alpha[:, o] - 0.5 * self.logdet - 0.5 * n * np.log(2 * np.pi)
                         for o in range(self.Y.shape[1])])

    def mean(self, Xs):
        return matern32(self.X, Xs, self.amp, self.inv_scale).T @ self.alpha          # (ns, m)

    def mean_var(self, Xs):
        Ks = matern32(self.X, Xs, self.amp, self.inv_scale)
        v = solve_triangular(self.L, Ks, lower=True)
        return Ks.T @ self.alpha, self.amp - np.sum(v * v, axis=0)

    def dmean_dx0(self, Xs, output=0):
        """d mean / d x_0 at every test point (first active dimension)."""
        d = (self.X[:, None, :] - Xs[None, :, :]) * self.inv_scale                     # x_i - x*
        r = np.sqrt(np.sum(d * d, axis=-1))
        # d/dx*_0 of A(1+sqrt3 r)exp(-sqrt3 r) = -3 A s0^2 (x*_0 - x_i0) exp(-sqrt3 r) = 3 A s0 d0 exp(-sqrt3 r)
        dk = 3.0 * self.amp * self.inv_scale[0] * d[:, :, 0] * np.exp(-SQRT3 * r)
        return dk.T @ self.alpha[:, output]


def neg_log_likelihood(theta, X, Y, sigma):
    """-sum_o log N(Y_o | 0, K(theta)), theta = [log_amp, log_scale_1..d], with its gradient."""
    X = np.asarray(X, float)
    Y = np.asarray(Y, float).reshape(len(X), -1)
    n, m = Y.shape
    with np.errstate(all='ignore'):
        amp, inv_scale = np.exp(theta[0]), np.exp(-theta[1:])
        diff = X[:, None, :] - X[None, :, :]
        sd = diff * inv_scale
        r = np.sqrt(np.sum(sd * sd, axis=-1))
        E = np.exp(-SQRT3 * r)
        Kf = amp * (1.0 + SQRT3 * r) * E
    if not np.all(np.isfinite(Kf)):     # a line-search probe far outside the sensible range: reject the step
        return 1e300, np.zeros_like(theta)
    K = Kf + sigma**2 * np.eye(n)
    try:
        c = cho_factor(K, lower=True)
    except np.linalg.LinAlgError:
        return 1e300, np.zeros_like(theta)
    alpha = cho_solve(c, Y)
    logdet = 2.0 * np.sum(np.log(np.diag(c[0])))
    f = 0.5 * np.sum(Y * alpha) + 0.5 * m * logdet + 0.5 * m * n * np.log(2 * np.pi)
    # dL/dtheta_k = -1/2 tr((alpha alpha^T - m K^-1) dK/dtheta_k)
    W = alpha @ alpha.T - m * cho_solve(c, np.eye(n))
    g = np.empty_like(theta)
    g[0] = -0.5 * np.sum(W * Kf)                                    # dK/dlog_amp = Kf
    # dk/dr = -3 A r exp(-sqrt3 r); r^2 = sum_j s_j^2 diff_j^2, s_j = exp(-l_j) -> dr/dl_j = -sd_j^2 / r
    for j in range(len(theta) - 1):
        dK = 3.0 * amp * E * sd[:, :, j]**2                         # = dk/dr * dr/dl_j (r cancels)
        g[1 + j] = -0.5 * np.sum(W * dK)
    return f, g


def train(X, Y, sigma, theta0):
    """Marginal-likelihood fit of (log_amp, log_scale) (gp.py:290-335: jaxopt.ScipyMinimize -> SciPy BFGS)."""
    res = minimize(neg_log_likelihood, np.asarray(theta0, float), args=(X, Y, sigma), jac=True, method='BFGS')
    return res.x, res.fun


def features(q, topo, extra):
    """(ncell, 7) feature matrix of all cells incl. ghosts, C order over (ix, iy) (gp.py:223-232)."""
    return np.vstack([q, topo[:3], extra]).reshape(7, -1).T


class OracleSurrogate:
    """One GP closure with fixed training data and hyper-parameters (no active learning)."""

    def __init__(self, kind, X7, Y13, Yerr13, theta, active_dims=None):
        """kind: 'press' | 'shear_x' | 'shear_y'; X7 (n,7) raw features, Y13/Yerr13 (n,13) raw outputs
        [p, tau_bot(6), tau_top(6)] and their standard errors (db.py:46-119)."""
        self.kind = kind
        X7, Y13, Yerr13 = np.asarray(X7, float), np.asarray(Y13, float), np.asarray(Yerr13, float)
        self.X_scale = np.maximum(np.max(np.abs(X7), axis=0), 1e-12)       # db.py:264-266
        Y_scale = np.maximum(np.max(np.abs(Y13), axis=0), 1e-12)
        if kind == 'press':
            self.dims = list(active_dims or [0, 3])                          # stress.py:498
            self.Yscale = Y_scale[0]                                         # stress.py:562-564
            Y = Y13[:, [0]] / self.Yscale
            yerr = np.mean((Yerr13 / Y_scale)[:, 0])                         # stress.py:566-569
        else:
            oi = 4 if kind == 'shear_x' else 3                               # stress.py:91
            self.dims = list(active_dims or ([0, 1, 3] if kind == 'shear_x' else [0, 2, 3]))
            cols = [oi + 1, oi + 7]
            self.Yscale = np.max(Y_scale[cols])                              # stress.py:239-242
            Y = Y13[:, cols] / self.Yscale
            yerr = np.mean(Yerr13[:, cols] / self.Yscale)                    # stress.py:255-258
        self.Xtrain = (X7 / self.X_scale)[:, self.dims]
        self.Ytrain, self.yerr = Y, float(yerr)
        self.theta = np.asarray(theta, float)
        self.fit = Fit(self.Xtrain, Y, np.exp(self.theta[0]), np.exp(-self.theta[1:]), self.yerr)
        self.variance = None

    @staticmethod
    def theta_init(Xtrain):
        return np.concatenate([[0.0], np.log(np.std(Xtrain, axis=0))])       # stress.py:281-284, 592-595

    def xtest(self, problem):
        return (features(problem.q, problem.topo, problem.extra) / self.X_scale)[:, self.dims]

    def init(self, problem):
        pass

    def predict(self, problem, predictor=False, compute_var=False):
        shape = problem.q.shape[1:]
        Xs = self.xtest(problem)
        if compute_var and predictor:
            m, v = self.fit.mean_var(Xs)
            self.variance = v.reshape(shape) * self.Yscale**2
        else:
            m = self.fit.mean(Xs)
        m = m * self.Yscale
        if self.kind == 'press':
            return m[:, 0].reshape(shape)
        return m[:, 0].reshape(shape), m[:, 1].reshape(shape)

    def v_sound(self, problem):
        g = self.fit.dmean_dx0(self.xtest(problem))
        return np.sqrt(g.max() * self.Yscale / self.X_scale[0])
