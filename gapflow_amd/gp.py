"""Gaussian-process surrogate closures: host-side control flow around the device GP kernels.

Mirrors (reference paths):
  GaPFlow/models/gp.py:46-603      GaussianProcessSurrogate (train / infer / active-learning loop)
  GaPFlow/models/stress.py:93-109, 195-287, 497-598   which features, outputs and scales each model uses
  GaPFlow/db.py:46-369             training database (arrays X (N,7), Y (N,13), Yerr (N,13), max-abs normalisers)
  GaPFlow/md/mock.py:32-107        Mock MD runner: fixed-form laws + noise

What runs where.  Per time step the posterior mean (every stage), the predictive variance and the GP
sound speed are HIP kernels (csrc/gp_kernels.hip) fed by a device-resident Cholesky factor (gpf_gp_set_model:
rocSOLVER dpotrf / dpotrs; GPF_USE_ROCSOLVER=0 or a missing rocSOLVER: the in-library blocked Cholesky).  Hyper-parameter training -- a handful of marginal-likelihood evaluations on <= a few
hundred points, run only when the database grows (gp.py:461-465) -- stays on the host with SciPy BFGS, the
optimiser the reference reaches through jaxopt.ScipyMinimize (gp.py:320-321).  Persistence of training data
in dtool datasets and the LAMMPS runners are out of scope; the database lives in memory.

Deliberate deviation: jax's PRNG streams (db.py:326-336, mock.py:82-88) cannot be reproduced without jax;
NumPy's default_rng with the same seeds is used instead.
"""
import contextlib
import ctypes as C
import io as _io
import os
from datetime import datetime

import numpy as np
from scipy.linalg import lapack
from scipy.optimize import minimize
from scipy.stats import qmc

from . import _lib

SQRT3 = np.sqrt(3.0)


# ---------------------------------------------------------------------------------------------
# training objective (host): -sum_o log N(Y_o | 0, K),  K = A (1 + sqrt3 r) exp(-sqrt3 r) + sigma^2 I
# ---------------------------------------------------------------------------------------------
def _single_threaded_blas():
    """NumPy and SciPy ship one OpenBLAS each; called alternately on few-hundred-point matrices their spinning thread
    pools fight over the cores (measured here: dpotrf of a 512^2 matrix 6 ms alone, 120 ms inside the objective).  One
    thread is faster at these sizes.  No-op without threadpoolctl."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1, user_api='blas')
    except ImportError:
        return contextlib.nullcontext()


class NegLogLikelihood:
    """theta -> (value, gradient) for one training set.  The squared coordinate differences do not depend on theta and are
    kept (one contiguous n x n array per input dimension), K^-1 comes from LAPACK dpotri: ~7x faster than the
    straightforward form at n = 512, which matters because every active-learning addition retrains."""

    def __init__(self, X, Y, sigma):
        X = np.asarray(X, float)
        self.Y = np.asarray(Y, float).reshape(len(X), -1)
        self.sigma2 = float(sigma)**2
        self.D2 = np.stack([np.subtract.outer(X[:, j], X[:, j])**2 for j in range(X.shape[1])])
        n, m = self.Y.shape
        self.const = 0.5 * m * n * np.log(2 * np.pi)

    def __call__(self, theta):
        Y, (n, m) = self.Y, self.Y.shape
        with np.errstate(all='ignore'):
            amp, inv_scale2 = np.exp(theta[0]), np.exp(-2.0 * theta[1:])
            r = np.sqrt(3.0 * np.tensordot(inv_scale2, self.D2, axes=1))         # sqrt3 * |scaled distance|
            E = np.exp(-r)
            Kf = amp * (1.0 + r) * E
        if not np.all(np.isfinite(Kf)):     # a line-search probe far outside the sensible range: reject the step
            return 1e300, np.zeros_like(theta)
        K = Kf.copy()
        K[np.diag_indices(n)] += self.sigma2
        c, info = lapack.dpotrf(K, lower=True, clean=False, overwrite_a=True)
        if info != 0:
            return 1e300, np.zeros_like(theta)
        alpha = lapack.dpotrs(c, Y, lower=True)[0]
        f = 0.5 * np.sum(Y * alpha) + m * np.sum(np.log(np.diag(c))) + self.const
        Kinv = lapack.dpotri(c, lower=True)[0]
        Kinv = np.tril(Kinv) + np.tril(Kinv, -1).T
        W = alpha @ alpha.T - m * Kinv          # dL/dtheta_k = -1/2 tr(W dK/dtheta_k)
        g = np.empty_like(theta)
        g[0] = -0.5 * np.sum(W * Kf)
        WE = W * E
        for j in range(len(theta) - 1):         # dK/dlog_scale_j = 3 A exp(-r) s_j^2 d_j^2
            g[1 + j] = -1.5 * amp * inv_scale2[j] * np.sum(WE * self.D2[j])
        return f, g


class DeviceNegLogLikelihood:
    """theta -> (value, gradient) evaluated by the library (gpf_gp_nll_*: kernel matrix, Cholesky, K^-1 and the gradient sums
    on the device; the optimiser stays on the host, as the reference's jaxopt.ScipyMinimize does).  Same arithmetic as
    NegLogLikelihood, which remains the statement of it that the tests compare with; use as a context manager."""

    def __init__(self, X, Y, sigma, device=0):
        self._lib = _lib.require_device()
        X = _lib.f64c(np.asarray(X, float))
        Y = _lib.f64c(np.asarray(Y, float).reshape(len(X), -1))
        self._h = C.c_void_p()
        _lib.check(self._lib.gpf_gp_nll_open(device, X.shape[0], X.shape[1], Y.shape[1], _lib.as_dp(X), _lib.as_dp(Y), float(sigma),
                                             C.byref(self._h)))
        self._nd = X.shape[1]
        self.not_positive_definite = 0      # probes at which the device factorisation found a non-positive pivot

    def __call__(self, theta):
        theta = _lib.f64c(np.asarray(theta, float))
        if not np.all(np.isfinite(theta)):
            return 1e300, np.zeros_like(theta)
        value, info = C.c_double(0.0), C.c_int(0)
        grad = np.zeros(1 + self._nd)
        _lib.check(self._lib.gpf_gp_nll_eval(self._h, _lib.as_dp(theta), C.byref(value), _lib.as_dp(grad), C.byref(info)))
        if info.value > 0:
            self.not_positive_definite += 1
        if info.value != 0 or not np.isfinite(value.value) or not np.all(np.isfinite(grad)):
            return 1e300, np.zeros_like(theta)         # a line-search probe far outside the sensible range: reject the step
        return value.value, grad

    def close(self):
        if self._h:
            self._lib.gpf_gp_nll_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass


def neg_log_likelihood(theta, X, Y, sigma):
    """Value and gradient w.r.t. theta = [log_amp, log_scale_1..d] (gp.py:307-318, 598-603)."""
    return NegLogLikelihood(X, Y, sigma)(theta)


# ---------------------------------------------------------------------------------------------
# Mock MD runner and in-memory database
# ---------------------------------------------------------------------------------------------
class Mock:
    """Stand-in for an MD run: the fixed-form laws at one state point plus a fixed pseudo-random offset
    (md/mock.py:81-107).  The laws are evaluated by the device closure kernel on a one-cell problem."""

    name = 'mock'
    is_mock = True

    def __init__(self, prop, geo, gp, device=0):
        self.device = device
        self.noise = (gp['press']['obs_stddev'] if gp['press_gp'] else 0.,
                      gp['shear']['obs_stddev'] if gp['shear_gp'] else 0.)
        self.prop, self.geo = prop, geo
        self.params = dict(prop)
        self._eval = None
        self._dtool_basepath = None     # set by Database.set_training_path; None: keep runs in memory only

    dtool_basepath = property(lambda self: self._dtool_basepath)

    def _write_dataset(self, X, Y, Ye, tag):
        """One directory per run holding a README.yml with the run's inputs and outputs (keys X / Y / Yerr / parameters, the
        ones md/base.py:128-188 records), which is all this package needs to reload a database.  It is NOT a dtool
        dataset: dtool persistence is out of scope and nothing of dtool's own on-disk
        format is imitated here."""
        import yaml
        now = datetime.now()
        name = f'{now.strftime("%Y%m%d_%H%M%S")}_{self.name}-{int(tag):03}'
        root = os.path.join(self._dtool_basepath, name)
        k = 0
        while os.path.exists(root):         # two runs within the same second and tag
            k += 1
            root = os.path.join(self._dtool_basepath, f'{name}_{k}')
        os.makedirs(root)
        plain = lambda a: [float(v) for v in np.asarray(a, float).ravel()]
        readme = {'parameters': {k_: (v if isinstance(v, (int, float, str, bool, dict, list)) else str(v)) for k_, v in self.params.items()},
                  'X': plain(X), 'Y': plain(Y), 'Yerr': plain(Ye)}
        with open(os.path.join(root, 'README.yml'), 'w') as f:
            yaml.safe_dump(readme, f)

    def _evaluator(self):
        if self._eval is None:
            from .problem import Problem
            grid = {'Nx': 1, 'Ny': 1, 'dx': 1., 'dy': 1., 'Lx': 1., 'Ly': 1., 'dim': 0}
            for side in ('xE', 'xW', 'yS', 'yN'):
                grid[f'bc_{side}_P'], grid[f'bc_{side}_D'], grid[f'bc_{side}_N'] = [True] * 3, [False] * 3, [False] * 3
            geo = dict(self.geo, type='inclined', hmin=1., hmax=1., flip=False)
            numerics = {'tol': 1e-6, 'max_it': 1, 'dt': 1e-12, 'adaptive': False, 'CFL': 0.5, 'MC_order': 1}
            prop = {k: v for k, v in self.prop.items() if k not in ('piezo', 'thinning')}
            prop.setdefault('elastic', {'enabled': False})
            # absent rho0: the EOS function's own default (pressure.py:73-76); it also seeds the evaluator's initial state
            prop.setdefault('rho0', _lib.EOS_DEFAULTS.get(prop['EOS'], {}).get('rho0', 1.0))
            self._eval = Problem({'output': '', 'write_freq': 1, 'use_tstamp': False, 'silent': True}, grid, numerics,
                                 prop, geo, device=self.device)
        return self._eval

    def run(self, X, tag=None):
        """X: 7 features [rho, jx, jy, h, dh/dx, dh/dy, extra] -> (Y (13,), Yerr (13,))."""
        ev = self._evaluator()
        X = np.asarray(X, float)
        ev.q[...] = X[:3, None, None]
        ev.topo.full[:3] = X[3:6, None, None]
        ev._upload_topo()
        ev._extra[...] = X[6]
        ev._upload(_lib.FIELD_EXTRA, ev._extra)
        ev._closures_stale = True
        p = ev._derived(_lib.FIELD_PRESSURE)[0, 1, 1]
        bot = ev._derived(_lib.FIELD_WALL_LOWER)[:, 1, 1]
        top = ev._derived(_lib.FIELD_WALL_UPPER)[:, 1, 1]
        rng = np.random.default_rng(123)                # one fixed draw per call, like jr.key(123) in mock.py:82
        noise_p, noise_s0, noise_s1 = rng.standard_normal(3) * np.array([self.noise[0], self.noise[1], self.noise[1]])
        Y = np.concatenate([[p + noise_p], bot + noise_s0, top + noise_s1])
        s = self.noise[1]
        Ye = np.array([self.noise[0], 0., 0., 0., s, s, 0., 0., 0., 0., s, s, 0.])
        if self._dtool_basepath is not None:
            self._write_dataset(X, Y, Ye, 0 if tag is None else tag)
        return Y, Ye


class Database:
    """Training data of all surrogates (db.py:46-369).  With a `dtool_path` every run is also kept as a plain directory
    with a README.yml below it and an existing directory is loaded on construction (the role of db.py:79-103; dtool
    itself is out of scope); without one the database lives in memory until a Problem gives it `<output>/train`
    (problem.py:158-161)."""

    def __init__(self, md, db, num_extra_features=1):
        self._md, self._db = md, db
        self._num_features = 6 + num_extra_features
        self._Xtrain = np.empty((0, self._num_features))
        self._Ytrain = np.empty((0, 13))
        self._Ytrain_err = np.empty((0, 13))
        self._X_scale = np.ones(self._num_features)
        self._Y_scale = np.ones(13)
        self.output_path = None
        self._training_path = None
        self._temporary_training_path = True
        path = db.get('dtool_path')
        if path is not None:
            self._temporary_training_path = False
            self.set_training_path(path)
            readmes = self.get_readme_list_local()
            if readmes:
                self.set_arrays([r['X'] for r in readmes], [r['Y'] for r in readmes], [r['Yerr'] for r in readmes])

    training_path = property(lambda self: self._training_path)

    def set_training_path(self, new_path, check_temporary=False):
        """db.py:236-262: where new runs are stored (also tells the MD runner)."""
        if check_temporary and not self._temporary_training_path:
            return
        os.makedirs(new_path, exist_ok=True)
        self._training_path = new_path
        self._md._dtool_basepath = new_path
        self._db['dtool_path'] = new_path

    def get_readme_list_local(self):
        """README.yml contents of the datasets below the training path, oldest first (db.py:193-209)."""
        import yaml
        out = []
        for name in sorted(os.listdir(self._training_path)):
            fn = os.path.join(self._training_path, name, 'README.yml')
            if os.path.isfile(fn):
                with open(fn) as f:
                    rm = yaml.safe_load(f)
                if isinstance(rm, dict) and all(k in rm for k in ('X', 'Y', 'Yerr')):
                    out.append(rm)
        print(f"Loading {len(out)} local datasets in '{self._training_path}'.")
        return out

    config = property(lambda self: self._db)
    md_config = property(lambda self: self._md.params)
    size = property(lambda self: self._Xtrain.shape[0])
    num_features = property(lambda self: self._num_features)
    has_mock_md = property(lambda self: self._md.is_mock)
    X_scale = property(lambda self: self._X_scale)
    Y_scale = property(lambda self: self._Y_scale)
    Xtrain = property(lambda self: self._Xtrain / self._X_scale)
    Ytrain = property(lambda self: self._Ytrain / self._Y_scale)
    Ytrain_err = property(lambda self: self._Ytrain_err / self._Y_scale)

    @staticmethod
    def _normalizer(x):
        return np.maximum(np.max(np.abs(x), axis=0), 1e-12)          # db.py:264-266

    def set_arrays(self, X, Y, Yerr):
        """Load a ready-made training set (replaces dtool persistence, db.py:83-103)."""
        self._Xtrain, self._Ytrain, self._Ytrain_err = (np.array(a, float) for a in (X, Y, Yerr))
        self._X_scale, self._Y_scale = self._normalizer(self._Xtrain), self._normalizer(self._Ytrain)

    def initialize(self, Xtest, dim=1):
        """Fill up to init_size points around the mean state (db.py:278-341)."""
        db = self._db
        nsample = db['init_size'] - self.size
        if nsample <= 0:
            return
        print(f"Database contains less than {db['init_size']} MD runs.")
        # `Xtest` is the (ncell, 7) feature table of all cells, or an object that serves its column means and rows on
        # demand (slab.py: the whole domain's table is never built per rank)
        lazy = hasattr(Xtest, 'column_mean')
        def dense_mean(k):
            # a uniform column (the initial field) has its own value as mean: exact, and therefore the same number on
            # every slab (np.mean of n equal values can be an ulp off, depending on n)
            col = Xtest[:, k]
            return float(col[0]) if (col == col[0]).all() else float(np.mean(col))
        mean = Xtest.column_mean if lazy else dense_mean
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
            m = int(np.ceil(np.log2(nsample)))
            samples = qmc.scale(qmc.Sobol(d=len(active), seed=rng).random_base2(m=m), lo, hi)
            nsample = samples.shape[0]
        choice = rng.choice(Xtest.shape[0], size=nsample, replace=False)
        if len(active) == 2:
            samples = np.hstack([samples, np.zeros((nsample, 1))])
        picked = Xtest.rows(choice) if lazy else Xtest[choice]
        self.add_data(np.column_stack([samples, picked[:, 3:]]))

    def add_data(self, Xnew):
        for X in np.atleast_2d(Xnew):
            Y, Ye = self._md.run(X, self.size + 1)
            self._Xtrain = np.vstack([self._Xtrain, X])
            self._Ytrain = np.vstack([self._Ytrain, Y])
            self._Ytrain_err = np.vstack([self._Ytrain_err, Ye])
            self._X_scale, self._Y_scale = self._normalizer(self._Xtrain), self._normalizer(self._Ytrain)


def make_database(input_dict, device=0):
    """problem.py:223-249: a `db` section without `md` attaches the Mock runner."""
    if input_dict.get('md') is not None:
        raise NotImplementedError("LAMMPS MD runners (GaPFlow/md/) are outside the scope of the MI355X hot path")
    gp = input_dict.get('gp')
    if gp is None:
        raise IOError("a `db` section needs a `gp` section")
    return Database(Mock(input_dict['properties'], input_dict['geometry'], gp, device), input_dict['db'])


# ---------------------------------------------------------------------------------------------
# one surrogate model (pressure, wall shear xz, wall shear yz)
# ---------------------------------------------------------------------------------------------
class Surrogate:
    """Host half of GaussianProcessSurrogate for one closure; the device half is gpf_gp_* ."""

    WHICH = {'zz': 0, 'xz': 1, 'yz': 2}

    def __init__(self, problem, name, cfg, database):
        self._p, self.name, self.database = problem, name, database
        self.which = self.WHICH[name]
        self.is_gp_model = True
        if name == 'zz':
            self.active_dims = list(cfg.get('active_dims', [0, 3]))                  # stress.py:498
            self._cols = [0]
        else:
            key, default = ('active_dims_x', [0, 1, 3]) if name == 'xz' else ('active_dims_y', [0, 2, 3])
            self.active_dims = list(cfg.get(key, default))                           # stress.py:94-95
            oi = 4 if name == 'xz' else 3
            self._cols = [oi + 1, oi + 7]                                            # stress.py:214-215
        self.atol, self.rtol = cfg['atol'], cfg['rtol']
        self.max_steps, self.pause_steps = cfg['max_steps'], cfg['pause_steps']
        self.use_active_learning = cfg['active_learning']
        self._step, self._pause, self.last_fit_train_size = 0, 0, 0
        self.optimise = True            # False: keep the initial hyper-parameters (timing runs, conditioning tests)
        self.theta = None
        self.params_init = None
        self.events = []                # ('train', step, reason, database size) / ('add', step, raw feature row): what the loop decided
        self._attached_scales = None    # (X_scale of the active dimensions, Yscale) the device model currently uses
        self.maximum_variance = np.inf
        self.variance_tol = 0.0
        self._var_valid = False
        self._var_computed = False      # a variance field exists on the device (possibly of an earlier state: gp.py:406-414)
        ref = datetime.now()
        self.cumtime_train = self.cumtime_infer = ref - ref
        self.history = {k: [] for k in ('step', 'database_size', 'variance', 'obs_stddev', 'maximum_variance', 'variance_tol')}
        for li in self.active_dims:
            self.history[f'lengthscale_{li}'] = []

    def __repr__(self):
        # the reference dumps the tinygp object here (problem.py:492); the same facts in plain text
        ls = ', '.join(f'{v:.6g}' for v in self.kernel_lengthscale) if self.theta is not None else '-'
        var = f'{self.kernel_variance:.6g}' if self.theta is not None else '-'
        return (f"GaussianProcess(kernel=Matern32, name={self.name}, active_dims={self.active_dims}, "
                f"n_train={self.last_fit_train_size}, variance={var}, lengthscales=[{ls}], "
                f"obs_stddev={self.obs_stddev if self.last_fit_train_size else '-'})")

    # -- training data views (stress.py:199-258, 546-569) ----------------------------------
    @property
    def Xtrain(self):
        return self.database.Xtrain[:self.last_fit_train_size, self.active_dims]

    @property
    def Yscale(self):
        return self.database.Y_scale[0] if self.name == 'zz' else np.max(self.database.Y_scale[self._cols])

    @property
    def Ytrain(self):
        return self.database._Ytrain[:self.last_fit_train_size][:, self._cols] / self.Yscale

    @property
    def Yerr(self):
        n = self.last_fit_train_size
        if self.name == 'zz':
            return float(np.mean(self.database.Ytrain_err[:n, 0]))
        return float(np.mean(self.database._Ytrain_err[:n][:, self._cols] / self.Yscale))

    obs_stddev = property(lambda self: self.Yerr)
    kernel_variance = property(lambda self: float(np.exp(self.theta[0])))
    kernel_lengthscale = property(lambda self: np.exp(-self.theta[1:]))      # tinygp Linear scale = exp(-log_scale)
    trusted = property(lambda self: self.maximum_variance < self.variance_tol)

    _say = staticmethod(print)          # the training / active-learning report (gp.py:296-300, 498-503)

    # -- train / attach ------------------------------------------------------------------------
    def train(self, reason=0, optimise=None):
        """gp.py:290-335.  optimise=False keeps theta (used by benchmarks with fixed hyper-parameters)."""
        self.last_fit_train_size = self.database.size
        self.events.append(('train', self._step, reason, self.database.size))
        say = self._say
        say('#' + 17 * '-' + f"GP TRAINING ({self.name.upper()})" + 17 * '-')
        say('# Timestep     :', self._step)
        say('# Reason       :', ['DB', 'AL'][reason])
        say('# Database size:', self.database.size)
        optimise = self.optimise if optimise is None else optimise
        X, Y, sigma = self.Xtrain, self.Ytrain, self.Yerr
        if self.params_init is None:
            # set ONCE, by init() (stress.py:281-284, 592-595): every later fit starts from the first training set's values
            self.params_init = np.concatenate([[0.0], np.log(np.std(X, axis=0))])
        theta0 = self.params_init
        if optimise or self.theta is None:
            if optimise:
                # objective and gradient on the device (GPF_GP_TRAIN=host: the NumPy / LAPACK statement of the same arithmetic)
                if os.environ.get('GPF_GP_TRAIN', 'device') == 'host':
                    with _single_threaded_blas():
                        res = minimize(NegLogLikelihood(X, Y, sigma), theta0, jac=True, method='BFGS')
                else:
                    dev = self._p._cfg.device if hasattr(self._p, '_cfg') else getattr(self._p, '_device', 0)
                    with DeviceNegLogLikelihood(X, Y, sigma, device=dev) as nll:
                        res = minimize(nll, theta0, jac=True, method='BFGS')
                        rejected = nll.not_positive_definite
                    # With a numerically singular K (observation noise ~1e-8 of the output scale) two Cholesky codes disagree on
                    # whether a probe is still positive definite; where the device factorisation gave up on probes that LAPACK
                    # may still factorise, the search can stop early: the host statement then continues from where it stopped.
                    if rejected:
                        with _single_threaded_blas():
                            host_nll = NegLogLikelihood(X, Y, sigma)
                            start = res.x if host_nll(res.x)[0] < 1e300 else theta0
                            res = minimize(host_nll, start, jac=True, method='BFGS')
                self.theta, obj = res.x, res.fun
            else:
                self.theta, obj = theta0, neg_log_likelihood(theta0, X, Y, sigma)[0]
            say(f'# Objective    : {obj:.5g}')
        self.attach()
        if self._step > 0:
            self.write()
        if reason == 0:
            say('#' + 50 * '-')

    def attach(self):
        """Factorise K on the device with the current data and hyper-parameters (gp.py:323)."""
        X, Y = _lib.f64c(self.Xtrain), _lib.f64c(self.Ytrain)
        dims = (C.c_int32 * len(self.active_dims))(*self.active_dims)
        xs = _lib.f64c(self.database.X_scale[self.active_dims])
        inv_scale = _lib.f64c(np.exp(-self.theta[1:]))
        p = self._p
        _lib.check(p._lib.gpf_gp_set_model(p._h, self.which, X.shape[0], X.shape[1], Y.shape[1], dims, _lib.as_dp(xs),
                                           _lib.as_dp(X), _lib.as_dp(Y), float(np.exp(self.theta[0])),
                                           _lib.as_dp(inv_scale), float(self.Yerr), float(self.Yscale)))
        p._closures_stale = True
        self._var_valid = False
        self._attached_scales = (np.array(xs), float(self.Yscale))

    def sync_scales(self):
        """The reference normalises test inputs and outputs with the database's CURRENT scales (the Xtest / Yscale
        properties, stress.py:195-242, 542-564) whatever the age of the fitted model: when another surrogate has added a point
        that moved a maximum, this model -- not refitted before the next predictor stage -- goes on with its old
        factorisation and the new scales.  Returns True if the device model's scales changed."""
        if self._attached_scales is None:
            return False
        xs, ys = _lib.f64c(self.database.X_scale[self.active_dims]), float(self.Yscale)
        if np.array_equal(xs, self._attached_scales[0]) and ys == self._attached_scales[1]:
            return False
        p = self._p
        _lib.check(p._lib.gpf_gp_set_scales(p._h, self.which, _lib.as_dp(xs), ys))
        self._attached_scales = (np.array(xs), ys)
        p._closures_stale = True
        self._var_valid = False
        return True

    def init(self):
        """stress.py:278-287, 586-598: first fit, then one inference WITH the variance (gp.py:390-414): maximum_variance and
        variance_tol exist from here on."""
        self.train(reason=0)
        self.compute_variance(on_open_step=False)

    def write(self):
        h = self.history
        h['step'].append(self._step)
        h['database_size'].append(self.database.size)
        h['variance'].append(self.kernel_variance)
        h['obs_stddev'].append(self.obs_stddev)
        h['maximum_variance'].append(self.maximum_variance)
        h['variance_tol'].append(self.variance_tol)
        for i, li in enumerate(self.active_dims):
            h[f'lengthscale_{li}'].append(self.kernel_lengthscale[i])

    # -- inference ---------------------------------------------------------------------------
    def compute_variance(self, on_open_step):
        """gp.py:406-410: variance field on the device, its maximum and the tolerance."""
        p = self._p
        mv = C.c_double(0)
        tic = datetime.now()
        _lib.check(p._lib.gpf_gp_variance(p._h, self.which, int(on_open_step), C.byref(mv)))
        self.cumtime_infer += datetime.now() - tic
        self.maximum_variance = mv.value
        self.variance_tol = max(self.atol * self.Yerr * self.Yscale, self.rtol * self.Yscale)**2
        self._var_valid = True
        self._var_computed = True

    @property
    def variance(self):
        field = {0: _lib.FIELD_PRESSURE_VAR, 1: _lib.FIELD_WALL_XZ_VAR, 2: _lib.FIELD_WALL_YZ_VAR}[self.which]
        return self._p._download(field, 1)[0]

    def _infer_mean_var(self):
        """Mean and variance of the current state (tests/test_inference.py:88-109 calls this directly)."""
        p = self._p
        p._sync_to_device()
        p._closures_stale = True
        self.compute_variance(on_open_step=False)
        if self.which == 0:
            mean = p._derived(_lib.FIELD_PRESSURE)[0]
        else:
            oi = 4 if self.which == 1 else 3
            mean = np.stack([p._derived(_lib.FIELD_WALL_LOWER)[oi], p._derived(_lib.FIELD_WALL_UPPER)[oi]])
        return mean, self.variance

    def _most_uncertain(self, features_of_cell):
        """Features of the cell with the largest predictive variance (gp.py:428-430)."""
        return features_of_cell(int(np.argmax(self.variance)))

    def stage(self, predictor, compute_var, features_of_cell):
        """The host side of predict() for one stage of an open step (gp.py:435-506).
        Returns True if the device model changed (the caller re-evaluates the stage's closures)."""
        changed = False
        if predictor:
            self._step += 1
            self._pause = max(-1, self._pause - 1)
            if self.last_fit_train_size < self.database.size:
                tic = datetime.now()
                self.train(reason=0)
                self.cumtime_train += datetime.now() - tic
                changed = True
        compute_var = self.use_active_learning or compute_var            # stress.py:353-354, 617-618
        if compute_var and predictor:
            self.compute_variance(on_open_step=True)
        if self.use_active_learning and predictor and self._pause < 0:
            counter = 0
            before = self.maximum_variance / self.variance_tol
            while not self.trusted and counter < self.max_steps:
                counter += 1
                Xnew = self._most_uncertain(features_of_cell)
                self.events.append(('add', self._step, np.array(Xnew, float)))
                self.database.add_data(Xnew[None, :])
                tic = datetime.now()
                self.train(reason=1)
                self.cumtime_train += datetime.now() - tic
                changed = True
                self.compute_variance(on_open_step=True)
                after = self.maximum_variance / self.variance_tol
                self._say(f"# AL {counter:2d}/{self.max_steps:2d}     : {before:.3f} --> {after:.3f}")
                self._say('#' + 50 * '-')
            if counter == self.max_steps:
                self._say("# Active learning loop missed uncertainty threshold")
                self._say(f"# Pause for {self.pause_steps} steps...")
                self._say('#' + 50 * '-')
                self._pause = self.pause_steps
        return changed


def attach_surrogates(problem, gp, database, cls=Surrogate):
    """Problem._select_gp_config (problem.py:643-660): press + shear-x in 1-D, + shear-y in 2-D."""
    models = {}
    if gp.get('press') is not None:
        models['zz'] = cls(problem, 'zz', gp['press'], database)
    if gp.get('shear') is not None:
        models['xz'] = cls(problem, 'xz', gp['shear'], database)
        if problem.grid['dim'] == 2:
            models['yz'] = cls(problem, 'yz', gp['shear'], database)
    return models


class SlabSurrogate(Surrogate):
    """A surrogate of a slab-decomposed problem.  Database, hyper-parameters and factorisation are replicated on
    every rank (same data, same deterministic host optimiser); only the two domain-wide decisions of active
    learning -- the largest variance and the cell it belongs to -- are exchanged."""

    def _say(self, *a, **k):
        if self._p.rank == 0:           # one training report per job, not per rank
            print(*a, **k)

    def compute_variance(self, on_open_step):
        super().compute_variance(on_open_step)
        self._local_maximum = self.maximum_variance
        self.maximum_variance = self._p.domain_max(self.maximum_variance)

    def _most_uncertain(self, features_of_cell):
        var = self.variance
        i = int(np.argmax(var))
        return self._p.features_of_domain_max(float(var.flat[i]), features_of_cell(i))
