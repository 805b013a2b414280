"""Blocked device Cholesky against LAPACK on the numerically singular kernel matrix a TRAINED pressure surrogate produces
(2-D slider, 256 Latin-hypercube points in a narrow box, noise 6e-5 of the output scale: cond(K) beyond 1/eps).
Usage: python tools/chol_probe.py"""
import contextlib, io, os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.linalg import lapack
from scipy.optimize import minimize
from bench import GP_YAML
from gapflow_amd import Problem, _lib
from gapflow_amd.gp import NegLogLikelihood
text = GP_YAML.format(n=256, nt=256).replace('obs_stddev: 100., active_learning: False', 'obs_stddev: 1.e5, active_learning: False')
with contextlib.redirect_stdout(io.StringIO()):
    prob = Problem.from_string(text)
    for m in prob._gp_models.values():
        m.optimise = False
    prob._pre_run()
m = prob._gp_models['zz']
X, Y, s = m.Xtrain, m.Ytrain, m.Yerr
theta0 = np.concatenate([[0.0], np.log(np.std(X, axis=0))])
th = minimize(NegLogLikelihood(X, Y, s), theta0, jac=True, method='BFGS').x
amp, inv = np.exp(th[0]), np.exp(-th[1:])
Z = X * inv
r = np.sqrt(3 * ((Z[:, None, :] - Z[None, :, :])**2).sum(-1))
K = amp * (1 + r) * np.exp(-r) + s**2 * np.eye(len(X))
w = np.linalg.eigvalsh(K)
print('theta', th, 'sigma', s, 'eigenvalues of K (fp64):', w[0], '...', w[-1])
c, info = lapack.dpotrf(K, lower=True)
print('LAPACK dpotrf info', info, 'min diag', np.diag(c).min() if info == 0 else None)
lib = _lib.require_device()
for env in ('', '1'):
    if env:
        os.environ['GPF_GP_UNBLOCKED_POTRF'] = env
    L = np.zeros_like(K); alpha = np.zeros((len(X), 1)); ld = C.c_double()
    rc = lib.gpf_gp_fit(0, len(X), X.shape[1], 1, _lib.as_dp(_lib.f64c(X)), _lib.as_dp(_lib.f64c(Y)), amp, _lib.as_dp(_lib.f64c(inv)), s,
                        _lib.as_dp(L), _lib.as_dp(alpha), C.byref(ld))
    print('device', 'unblocked' if env else 'blocked', 'rc', rc, lib.gpf_last_error().decode() if rc else '',
          'min diag', np.diag(L).min() if rc == 0 else None, 'max |L - LAPACK|', np.abs(L - np.tril(c)).max() if rc == 0 and info == 0 else None)
