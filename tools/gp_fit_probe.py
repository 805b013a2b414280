import sys, os, time, ctypes as C, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
faulthandler.dump_traceback_later(40, exit=False)
from gapflow_amd import _lib
lib = _lib.require_device()
print('loaded', flush=True)
rng = np.random.default_rng(3)
for n, d, m in ((64, 2, 1), (200, 3, 2), (512, 3, 2), (2, 2, 1)):
    X = rng.uniform(0.5, 1.0, (n, d)); Y = rng.standard_normal((n, m))
    if n == 2: X[1] = X[0]
    inv = np.ones(d); L = np.empty((n, n)); al = np.empty((n, m)); ld = C.c_double(0)
    t = time.time()
    print('fit', n, d, m, '...', flush=True)
    rc = lib.gpf_gp_fit(0, n, d, m, _lib.as_dp(_lib.f64c(X)), _lib.as_dp(_lib.f64c(Y)), 1.3, _lib.as_dp(inv), 0.05 if n > 2 else 0.0, _lib.as_dp(L), _lib.as_dp(al), C.byref(ld))
    print('   rc', rc, lib.gpf_last_error().decode()[:100], 'logdet', ld.value, 'time %.2f' % (time.time() - t), flush=True)
print('done', flush=True)
