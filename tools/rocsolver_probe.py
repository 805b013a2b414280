import ctypes as C, numpy as np, sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
t0 = time.time()
if os.environ.get('PRELOAD_ROCSOLVER'):
    C.CDLL('librocblas.so', mode=C.RTLD_GLOBAL); C.CDLL('librocsolver.so.0', mode=C.RTLD_GLOBAL)
    print('rocsolver preloaded before any HIP call', time.time() - t0, flush=True)
from gapflow_amd import _lib
lib = _lib.require_device()
rng = np.random.default_rng(4)
n, d, m = 300, 3, 2
X = rng.uniform(0.5, 1.0, (n, d)); Y = rng.standard_normal((n, m))
Xc, Yc, sc = _lib.f64c(X), _lib.f64c(Y), _lib.f64c(np.array([2.0, 0.8, 1.5]))
L, alpha, logdet = np.empty((n, n)), np.empty((n, m)), C.c_double(0)
print('calling gpf_gp_fit', time.time() - t0, flush=True)
_lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), 1.2, _lib.as_dp(sc), 0.05, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
print('first fit done', time.time() - t0, flush=True)
t1 = time.time()
_lib.check(lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(Xc), _lib.as_dp(Yc), 1.2, _lib.as_dp(sc), 0.05, _lib.as_dp(L), _lib.as_dp(alpha), C.byref(logdet)))
print('second fit', time.time() - t1, flush=True)
