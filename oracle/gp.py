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


# ---------------------------------------------------------------------------------------------
# Active learning: the surrogate's train / infer / add-a-point loop, the database it
# extends and the Mock MD runner that answers.  PARITY UNPINNED like the rest of this file (the reference
# holds no numeric fixture of an active-learning run); what IS restated line by line is the control flow.
#
#   GaPFlow/models/gp.py:290-335    _train: _last_fit_train_size = database.size, optimise from params_init
#                                   (set ONCE, in init(): stress.py:281-284, 592-595), _cache = None
#   gp.py:390-414                   _infer: with compute_var the variance field, its maximum and
#                                   variance_tol = max(atol Yerr Yscale, rtol Yscale)^2; otherwise the mean alone and the
#                                   variance field of an EARLIER evaluation stays
#   gp.py:419-430                   _active_learning: imax = argmax(var) over all cells incl. ghosts (first hit),
#                                   database.add_data(_Xtest[imax])
#   gp.py:435-506                   predict: predictor stage only: _step += 1, _pause = max(-1, _pause - 1), retrain when
#                                   the database has grown; variance only if compute_var AND predictor; the loop
#                                   `while not trusted and counter < max_steps`; _pause = pause_steps when it ran out
#   stress.py:353-354, 617-618      compute_var handed to predict = use_active_learning or compute_var
#   problem.py:530                  compute_var = one step before an output step
#   db.py:264-266, 278-369          max-abs normalisers recomputed after every added point; initialize(); add_data()
#   md/mock.py:81-107               Mock run = fixed-form laws + three fixed normal draws
#
# Quirks kept because the product has to match them:
#   * Xtrain is normalised by the database's CURRENT X_scale and the test inputs likewise, Yscale is the CURRENT one
#     (properties, stress.py:195-258, 542-569), while the factorised model and the cached alpha (gp.py:343-349) are those
#     of the last fit.  Between a growth of the database by ANOTHER model and this model's next fit (one corrector stage)
#     the mean is therefore Ks(X* / X_scale_now)^T alpha_fit * Yscale_now.
#   * the GP sound speed goes through gp.predict(self.Ytrain, x) (stress.py:588, 533-535), i.e. a fresh
#     alpha = K_fit^-1 (Y_raw / Yscale_now): times Yscale_now the output scale cancels, the input scale does not.
# Deviation shared with the product: jax's PRNG streams -> NumPy default_rng with the same seeds.
# ---------------------------------------------------------------------------------------------
class OracleMock:
    def __init__(self, prop, geo, gp):
        self.noise = (gp['press']['obs_stddev'] if gp['press_gp'] else 0., gp['shear']['obs_stddev'] if gp['shear_gp'] else 0.)
        self.prop, self.geo = prop, geo

    def run(self, X):
        from . import closures as cl
        X = np.asarray(X, float)
        n = np.random.default_rng(123).standard_normal(3) * np.array([self.noise[0], self.noise[1], self.noise[1]])
        q, h = X[:3, None], X[3:6, None]
        U, V, eta, zeta = self.geo['U'], self.geo['V'], self.prop['shear'], self.prop['bulk']
        bot = cl.stress_bottom(q, h, U, V, eta, zeta, X[6:7])[:, 0] + n[1]
        top = cl.stress_top(q, h, U, V, eta, zeta, X[6:7])[:, 0] + n[2]
        p = cl.eos_pressure(X[0:1], self.prop)[0] + n[0]
        s = self.noise[1]
        return np.concatenate([[p], bot, top]), np.array([self.noise[0], 0., 0., 0., s, s, 0., 0., 0., 0., s, s, 0.])


class OracleDatabase:
    def __init__(self, md, db, num_features=7):
        self._md, self._db = md, db
        self._Xtrain, self._Ytrain, self._Ytrain_err = np.empty((0, num_features)), np.empty((0, 13)), np.empty((0, 13))
        self.X_scale, self.Y_scale = np.ones(num_features), np.ones(13)
        self.added = []                 # (raw feature row) of every point added after initialize(), in order

    size = property(lambda self: self._Xtrain.shape[0])
    Xtrain = property(lambda self: self._Xtrain / self.X_scale)
    Ytrain_err = property(lambda self: self._Ytrain_err / self.Y_scale)

    @staticmethod
    def _normalizer(x):
        return np.maximum(np.max(np.abs(x), axis=0), 1e-12)

    def add_data(self, Xnew, record=True):
        for X in np.atleast_2d(Xnew):
            Y, Ye = self._md.run(X)
            self._Xtrain = np.vstack([self._Xtrain, X])
            self._Ytrain = np.vstack([self._Ytrain, Y])
            self._Ytrain_err = np.vstack([self._Ytrain_err, Ye])
            self.X_scale, self.Y_scale = self._normalizer(self._Xtrain), self._normalizer(self._Ytrain)
            if record:
                self.added.append(np.array(X, float))

    def initialize(self, Xtest, dim):
        from scipy.stats import qmc
        db = self._db
        nsample = db['init_size'] - self.size
        if nsample <= 0:
            return
        mean = lambda k: float(Xtest[0, k]) if (Xtest[:, k] == Xtest[0, k]).all() else float(np.mean(Xtest[:, k]))
        if dim == 1:
            flux, active = mean(1), [0, 1]
        else:
            flux, active = np.hypot(mean(1), mean(2)), [0, 1, 2]
        rho, w = mean(0), db['init_width']
        lo = np.array([(1.0 - w) * rho, 0.5 * flux, -0.5 * flux])[active]
        hi = np.array([(1.0 + w) * rho, 1.5 * flux, 0.5 * flux])[active]
        rng = np.random.default_rng(db['init_seed'])
        if db['init_method'] == 'rand':
            samples = rng.uniform(lo, hi, size=(nsample, len(active)))
        elif db['init_method'] == 'lhc':
            samples = qmc.scale(qmc.LatinHypercube(d=len(active), seed=rng).random(n=nsample), lo, hi)
        else:
            samples = qmc.scale(qmc.Sobol(d=len(active), seed=rng).random_base2(m=int(np.ceil(np.log2(nsample)))), lo, hi)
            nsample = samples.shape[0]
        choice = rng.choice(Xtest.shape[0], size=nsample, replace=False)
        if len(active) == 2:
            samples = np.hstack([samples, np.zeros((nsample, 1))])
        self.add_data(np.column_stack([samples, Xtest[choice, 3:]]), record=False)


class OracleGP:
    """GaussianProcessSurrogate + the Pressure / WallStress properties for one closure; same interface towards
    OracleProblem as OracleSurrogate.  `events` lists what happened, for the parity test:
    ('train', step, reason, database size), ('add', step, cell index, raw features)."""

    def __init__(self, kind, cfg, database, optimise=True):
        self.kind, self.database, self.optimise = kind, database, optimise
        if kind == 'press':
            self.dims, self.cols = list(cfg.get('active_dims', [0, 3])), [0]
        else:
            oi = 4 if kind == 'shear_x' else 3
            key, default = ('active_dims_x', [0, 1, 3]) if kind == 'shear_x' else ('active_dims_y', [0, 2, 3])
            self.dims, self.cols = list(cfg.get(key, default)), [oi + 1, oi + 7]
        self.atol, self.rtol = cfg['atol'], cfg['rtol']
        self.max_steps, self.pause_steps, self.use_active_learning = cfg['max_steps'], cfg['pause_steps'], cfg['active_learning']
        self._step, self._pause, self.last_fit_train_size = 0, 0, 0
        self.variance = None            # the stored variance field (raw units), possibly of an earlier state
        self.maximum_variance, self.variance_tol = None, None
        self.events = []
        self.fit = None

    # -- stress.py:195-258, 542-569 ----------------------------------------------------------------
    @property
    def Xtrain(self):
        return self.database.Xtrain[:, self.dims]          # all rows: equal to [:last_fit_train_size] whenever it is used to fit

    @property
    def Yscale(self):
        return self.database.Y_scale[0] if self.kind == 'press' else np.max(self.database.Y_scale[self.cols])

    @property
    def Ytrain(self):
        return self.database._Ytrain[:self.last_fit_train_size][:, self.cols] / self.Yscale

    @property
    def Yerr(self):
        n = self.last_fit_train_size
        if self.kind == 'press':
            return float(np.mean(self.database.Ytrain_err[:n, 0]))
        return float(np.mean(self.database._Ytrain_err[:n][:, self.cols] / self.Yscale))

    trusted = property(lambda self: self.maximum_variance < self.variance_tol)

    def xtest(self, problem):
        return (features(problem.q, problem.topo, problem.extra) / self.database.X_scale)[:, self.dims]

    # -- gp.py:290-335 -------------------------------------------------------------------------------
    def _train(self, reason):
        self.last_fit_train_size = self.database.size
        X, Y, sigma = self.Xtrain, self.Ytrain, self.Yerr
        self.theta = train(X, Y, sigma, self.params_init)[0] if self.optimise else np.array(self.params_init)
        self.fit = Fit(X, Y, np.exp(self.theta[0]), np.exp(-self.theta[1:]), sigma)
        self.Yscale_fit = self.Yscale
        self.events.append(('train', self._step, reason, self.database.size))

    def init(self, problem):
        # Problem._pre_run: init_database (problem.py:418-420) then init (stress.py:278-287, 586-598)
        self.database.initialize(features(problem.q, problem.topo, problem.extra), problem.grid['dim'])
        self.params_init = np.concatenate([[0.0], np.log(np.std(self.Xtrain, axis=0))])
        self._train(0)
        self._infer(problem, True)

    # -- gp.py:337-414 -------------------------------------------------------------------------------
    def _infer(self, problem, compute_var):
        Xs = self.xtest(problem)
        shape = problem.q.shape[1:]
        if compute_var:
            m, v = self.fit.mean_var(Xs)
            self.variance = v.reshape(shape) * self.Yscale**2
            self.maximum_variance = self.variance.max()
            self.variance_tol = max(self.atol * self.Yerr * self.Yscale, self.rtol * self.Yscale)**2
        else:
            m = self.fit.mean(Xs)
        return (m * self.Yscale).T.reshape((-1,) + shape)

    # -- gp.py:419-506 -------------------------------------------------------------------------------
    def predict(self, problem, predictor=False, compute_var=False):
        compute_var = self.use_active_learning or compute_var              # stress.py:353-354, 617-618
        if predictor:
            self._step += 1
            self._pause = max(-1, self._pause - 1)
            if self.last_fit_train_size < self.database.size:
                self._train(0)
        m = self._infer(problem, compute_var and predictor)
        if self.use_active_learning and predictor and self._pause < 0:
            counter = 0
            while not self.trusted and counter < self.max_steps:
                counter += 1
                imax = int(np.argmax(self.variance))                        # first hit, all cells incl. ghosts
                Xnew = features(problem.q, problem.topo, problem.extra)[imax]
                self.events.append(('add', self._step, imax, Xnew.copy()))
                self.database.add_data(Xnew[None, :])
                self._train(1)
                m = self._infer(problem, True)
            if counter == self.max_steps:
                self._pause = self.pause_steps
        return m[0] if self.kind == 'press' else (m[0], m[1])

    def v_sound(self, problem):
        # stress.py:533-535 with self.eos = gp.predict(self.Ytrain, .): alpha = K_fit^-1 (Y_raw / Yscale_now)
        g = self.fit.dmean_dx0(self.xtest(problem)) * (self.Yscale_fit / self.Yscale)
        return np.sqrt(g.max() * self.Yscale / self.database.X_scale[0])


def attach(problem, input_dict, optimise=True):
    """Problem.__init__ / _select_gp_config (problem.py:223-249, 643-660) for the oracle: a Mock-backed database and
    press + shear-x (+ shear-y in 2-D) surrogates, plugged into `problem.gp_models`."""
    gp, db = input_dict['gp'], input_dict['db']
    database = OracleDatabase(OracleMock(input_dict['properties'], input_dict['geometry'], gp), db)
    models = {}
    if gp.get('press') is not None:
        models['press'] = OracleGP('press', gp['press'], database, optimise)
    if gp.get('shear') is not None:
        models['shear_x'] = OracleGP('shear_x', gp['shear'], database, optimise)
        if problem.grid['dim'] == 2:
            models['shear_y'] = OracleGP('shear_y', gp['shear'], database, optimise)
    problem.gp_models = models
    return database, models
